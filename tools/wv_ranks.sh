# Dev tool: one rank vs four ranks that share the card (tools/debug_ranks.py), concurrently and with the predict stages one at a time
mkdir -p /tmp/dr; export LAYERS=4
for L in 1 8; do
  export LANES=$L
  python tools/debug_ranks.py /tmp/dr 2>&1 | grep sha1 > /dev/null
  for i in 1 2 3; do echo "-- 4 ranks concurrent, $L lane(s) each, run $i"; python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 2968$i tools/debug_ranks.py /tmp/dr 2>&1 | grep "affs\|largest"; done
done
