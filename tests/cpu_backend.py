"""TEST INFRASTRUCTURE: CPU stand-in for deepmod_amd.stream.HipBackend (oracle classifier + numpy counters), so that
the streaming engine's host logic and its multi-rank merge can be exercised without a GPU.  Never imported by the product."""
import numpy as np

from oracle import oracle_np


class CpuSummary:
    def __init__(self, length):
        self.length = int(length)
        self.counts = np.zeros(3 * self.length, np.int32)        # touch | cov | mod, like dm_summary's device block

    def grow(self, new_length):
        new_length = int(new_length)
        if new_length <= self.length:
            return
        nc = np.zeros(3 * new_length, np.int32)
        for k in range(3):
            nc[k * new_length:k * new_length + self.length] = self.counts[k * self.length:(k + 1) * self.length]
        self.counts, self.length = nc, new_length

    def add(self, pos, flags):
        L = self.length
        oracle_np.summary_add_c(self.counts[:L], self.counts[L:2 * L], self.counts[2 * L:], pos, flags)

    def fetch(self):
        L = self.length
        return self.counts[:L].copy(), self.counts[L:2 * L].copy(), self.counts[2 * L:].copy()

    def sync(self):
        pass

    def take_slice(self, rank, world):
        """dm_summary_reduce_scatter's slice rule (the counters must already hold the all-rank sums: the transport stand-in of
        the test all-reduces them): chunk = ceil(length / world), this rank owns [rank * chunk, min(length, (rank + 1) * chunk))."""
        chunk = -(-self.length // world)
        first = min(self.length, rank * chunk)
        self._slice = (first, min(self.length, (rank + 1) * chunk) - first)
        return self._slice

    def fetch_slice(self):
        L, (first, count) = self.length, self._slice
        return tuple(self.counts[k * L + first:k * L + first + count].copy() for k in range(3))

    def close(self):
        pass


class OracleBackend:
    def __init__(self, weights):
        self.w = weights

    def new_summary(self, length):
        return CpuSummary(length)

    def submit(self, pb, summaries):
        try:
            self._submit(pb, summaries)
        finally:
            if pb.on_done is not None:          # the backend's contract: the batch's own arrays are free on return
                pb.on_done()
                pb.on_done = None

    def _submit(self, pb, summaries):
        if pb.n_rows == 0:
            return
        R = pb.n_rows
        from numpy.lib.stride_tricks import sliding_window_view
        win = sliding_window_view(pb.rows, (21, 7))[:, 0]                  # window centred on row r + 10
        if getattr(pb, 'sel', None) is not None:
            # compact form (stream.Prepared.sel): only the windows centred on a base of interest are classified
            S = len(pb.sel)
            cls = oracle_np.predict_windows_c(self.w, np.ascontiguousarray(win[pb.sel - 10]))[1].astype(np.uint8) if S else np.zeros(0, np.uint8)
            for g in pb.groups:
                c, s, lo, hi, xlo, xhi, slo, shi = g
                summ = summaries(c, s, pb.contig_len.get(c, 0))
                if shi > slo:
                    summ.add(pb.pos[slo:shi], ((pb.flags[slo:shi] & 3) | (cls[slo:shi] << 2)).astype(np.uint8))
                if xhi > xlo:
                    summ.add(pb.pos[S + xlo:S + xhi], pb.flags[S + xlo:S + xhi])
            return
        cls = np.zeros(R, np.uint8)
        cls[10:R - 10] = oracle_np.predict_windows_c(self.w, np.ascontiguousarray(win))[1]
        for (c, s, lo, hi, xlo, xhi) in pb.groups:
            summ = summaries(c, s, pb.contig_len.get(c, 0))
            fl = (pb.flags[lo:hi] & 3) | (cls[lo:hi] << 2)
            summ.add(pb.pos[lo:hi], fl.astype(np.uint8))
            if xhi > xlo:
                summ.add(pb.pos[R + xlo:R + xhi], pb.flags[R + xlo:R + xhi])

    def sync(self):
        pass

    def close(self):
        pass
