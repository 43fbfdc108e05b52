cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -8
import os, sys, json, time
sys.path.insert(0, os.getcwd())
import torch
from seeme_amd import cli
out = cli.train_main(["--cfg", "configs/config_mld_gimo.yaml", "--batch_size", "64", "--nodebug", "--folder", "/tmp/exp_gimo",
                      "--epochs", "2", "--iters_per_epoch", "6"])
print(json.dumps({k: v for k, v in out.items() if k != "folder"}))
res = cli.test_main(["--cfg", "configs/config_mld_gimo.yaml", "--batch_size", "32", "--folder", "/tmp/exp_gimo", "--test_batches", "2",
                     "--checkpoint", os.path.join(out["checkpoints"], "epoch=1.ckpt")])
print({k: v for k, v in res.items() if k.endswith("/mean")})
PY
