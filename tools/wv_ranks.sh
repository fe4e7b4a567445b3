# Dev tool: runtime switches against the two-engines-side-by-side corruption (tools/debug_two_streams.py)
for cfg in "AMD_OPT_FLUSH=0" "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1" "GPU_MAX_HW_QUEUES=2" "GPU_MAX_HW_QUEUES=1" "HSA_ENABLE_SDMA=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 ROC_SIGNAL_POOL_SIZE=4096"; do
  echo "== $cfg"; env $cfg timeout -k 10 120 python tools/debug_two_streams.py 2>&1 | grep "differ\|rror" | tail -2
done
