"""GPU probe: cost of one in-launch all-reduce of a 256-float vector among the C workgroups (CUs) of a cluster, data-tagged
8-byte granules (probes/xchg_test.hip), with and without a weight stream on the same CUs.  Prints one JSON line per case."""
import ctypes as C
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seeme_amd import _lib as L
from probes import probe_lib

lib = probe_lib.lib()
f = lib.seeme_debug_xchg
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
iters = 4000
wbuf = torch.randint(0, 2 ** 31 - 1, (9 * 1024 * 1024 // 4,), dtype=torch.int32, device=dev)   # 9 MB "weight image"
for Cc in (8, 4, 2):
    for clusters in (256 // Cc, 8):
        if (clusters * Cc) % (8 * Cc) != 0:
            continue
        for mode in (0, 1):
            for stream_ld, kind in ((0, 0), (8, 0), (32, 0), (0, 1), (8, 1)):
              if kind == 1 and mode == 1:
                continue
              if True:
                ts = []
                errs = None
                for rep in range(4):
                    gran = torch.zeros(clusters * 2 * Cc * 256, dtype=torch.int64, device=dev)
                    err = torch.zeros(4, dtype=torch.int32, device=dev)
                    out = torch.zeros(2 * clusters * Cc, device=dev)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    probe_lib.check(f(gran.data_ptr(), wbuf.data_ptr(), wbuf.numel() // 4, Cc, clusters, mode, iters, stream_ld,
                                      err.data_ptr(), out.data_ptr(), kind, L.current_stream()))
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                    errs = err.tolist()
                us = sorted(ts)[1]
                g = clusters * Cc
                xcc = out[g:].to(torch.int32).tolist()
                same = all(len({xcc[b] for b in range(g) if b % 8 == x}) == 1 for x in range(8))   # blocks b, b + 8 on one XCD?
                distinct = len({xcc[x] for x in range(8)}) == 8
                print(json.dumps({"C": Cc, "clusters": clusters, "placement": "same_xcd" if mode == 0 else "across_xcds",
                                  "stream_KiB_per_cu_per_iter": stream_ld * 8, "us_per_exchange": round(us / iters, 3),
                                  "store": "sc1" if kind == 0 else "plain", "timeout": errs[0], "bad_sums": errs[1], "max_spins": errs[2], "b_mod_8_shares_xcd": same, "b_0_7_distinct_xcds": distinct}), flush=True)
