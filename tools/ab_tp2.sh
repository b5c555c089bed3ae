# two ranks on ONE GPU (gloo handles + IPC data): decode step with the dense-partial / slab all-reduce routes, with and without ACT-mode down
p=29610
for cfg in "0 1" "1 1" "1 0" "0 1" "1 0"; do
set -- $cfg; p=$((p+1))
EXL3_HIP_AR_FROM_SLABS=$1 EXL3_HIP_ACT_IN_GEMV=$2 EXL3_HIP_TP_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $p bench.py --gpus 2 --no-extra --no-cpu --no-prefill --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('slabs=$1 act=$2', d['value'], d['ms_per_step'], d.get('repeat_ms_per_step'))"
done
