#!/bin/bash
# Dev tool (GPU box): the raw-container command on a long input (tools/raw_profile.py, the generated reads 30 x by links: 3e9 base-positions, ~17 s of detect)
# while the device's memory in use and the resident set of the command's processes are sampled once a second - flat lines = no growth with the length of a run.
#   bash tools/soak_raw.sh [repeat]  > gpurun_out/r06/soak_raw.txt
REP=${1:-30}
cd "$(dirname "$0")/.."
BUS=$(python -c "from deepmod_amd import _lib; print(_lib.pci_bus_id(0))")     # the device the command runs on (other cards of the node belong to other jobs)
echo "device 0 = $BUS"
( while true; do
    v=$(cat /sys/bus/pci/devices/$BUS/mem_info_vram_used 2>/dev/null || echo 0)
    r=$(ps -eo rss,args | grep "DeepMod.py detect" | grep -v grep | awk '{s+=$1} END {print s+0}')
    echo "sample t=$(date +%s) vram_used_MB=$((v/1048576)) rss_of_detect_processes_MB=$((r/1024))"
    sleep 1
  done ) &
SAMPLER=$!
python tools/raw_profile.py 20000 "$REP" 4 2>&1 | grep -E "generated|Streaming detect|whole command|BED files"
kill $SAMPLER
