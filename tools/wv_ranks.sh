F="--volume 512 --steps 4 --warmup 1 --no-cpu-baseline --no-modes --no-train --no-drivers"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d["whole_volume"]; print(w["n_gpus"], w["fragments"], w["scored_edges"], w["segments"], round(w["seconds"],2))'
for i in 1 2; do python bench.py --gpus 1 $F 2>/dev/null | python -c "$P"; done
for i in 1 2 3; do python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2966$i bench.py --gpus 2 --backend gloo $F 2>/dev/null | python -c "$P"; done
for i in 1 2; do python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 2967$i bench.py --gpus 4 --backend gloo $F 2>/dev/null | python -c "$P"; done
