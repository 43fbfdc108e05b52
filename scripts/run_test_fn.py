"""Debug: call one test function of tests/ directly (no pytest): python scripts/run_test_fn.py test_gpu_flows test_name"""
import importlib, os, sys
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, repo); sys.path.insert(0, os.path.join(repo, "tests"))
import torch
m = importlib.import_module(sys.argv[1])
getattr(m, sys.argv[2])(torch.device("cuda:0"))
print("OK", sys.argv[2])
