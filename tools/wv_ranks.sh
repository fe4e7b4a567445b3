# Dev tool: one rank vs four ranks that share the card (tools/debug_ranks.py), concurrently and with the predict stages one at a time
mkdir -p /tmp/dr; export LAYERS=4
python tools/debug_ranks.py /tmp/dr 2>&1 | grep sha1 > /dev/null
for i in 1 2; do echo "-- 4 ranks concurrent $i"; python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 2968$i tools/debug_ranks.py /tmp/dr 2>&1 | grep "equal\|DIFFER\|largest"; done
echo "-- 4 ranks, predict one rank at a time"; SERIAL=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29691 tools/debug_ranks.py /tmp/dr 2>&1 | grep "equal\|DIFFER"
