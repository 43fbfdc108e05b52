# two ranks of bench.py on ONE card over gloo: exercises the rendezvous / barrier / max-over-ranks path
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
SEEME_BENCH_BACKEND=gloo SEEME_BENCH_DEVICE=0 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/dist2.json 2> gpurun_out/dist2.err
echo "rc=$?"; tail -2 gpurun_out/dist2.err; cat gpurun_out/dist2.json
