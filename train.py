"""python train.py --cfg configs/config_mld_egobody.yaml --nodebug   (torchrun --nproc-per-node N train.py ... for N GPUs)
Lightning-free equivalent of the reference's train.py on the MI355X path; see seeme_amd/cli.py."""
from seeme_amd.cli import train_main

if __name__ == "__main__":
    train_main()
