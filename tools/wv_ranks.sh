mkdir -p /tmp/dr; export LAYERS=4
for cfg in "BSMI_WINO4=0" "BSMI_WINO=0" "BSMI_WINO=0 BSMI_H16=0" "BSMI_WINO=0 BSMI_H16=0 BSMI_FUSED_FIRST=0 BSMI_USE_BOX=0" "BSMI_WINO=0 BSMI_H16=0 BSMI_SK_GRID=0"; do
  echo "== $cfg"
  env $cfg python tools/debug_ranks.py /tmp/dr 2>&1 | grep sha1 > /dev/null
  for i in 1 2; do env $cfg python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 2968$i tools/debug_ranks.py /tmp/dr 2>&1 | grep "affs"; done
done
