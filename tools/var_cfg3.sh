# cfg3_highcard_columns (4 columns x 45 M rows, ~36 M distinct ids) with and without the batched ordering
cd $GRAFT_REPO_ROOT; export NVT_READBACK_TIMEOUT=120
run() { env $2 timeout 300 python -c "
import torch, tempfile, bench
x = bench.extra_cfg3(torch.device('cuda', 0), tempfile.mkdtemp(), 45_000_000)
print('$1', x['ms_per_step'], x.get('gpu_busy_ms'), x.get('per_kernel_ms'))" 2>/dev/null | tail -1; }
run base NVT_X=1; run nobatch NVT_NO_ORDER_BATCH=1; run base NVT_X=1; run nobatch NVT_NO_ORDER_BATCH=1
