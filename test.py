"""python test.py --cfg configs/config_mld_egobody.yaml --checkpoint <FOLDER>/mld/<NAME>/checkpoints/epoch=N.ckpt
Lightning-free equivalent of the reference's test.py on the MI355X path; see seeme_amd/cli.py."""
from seeme_amd.cli import test_main

if __name__ == "__main__":
    test_main()
