"""Ad-hoc: forward kernel time in evaluation mode (library HIP-event hooks) for both mappings / several pack sizes.
usage: python tools/gpu_time_eval.py [case] [batch] [K]"""
import sys, os, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import opf_graph_neural_solver_amd as amd
case = int(sys.argv[1]) if len(sys.argv) > 1 else 118
bt = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
K = int(sys.argv[3]) if len(sys.argv) > 3 else 4
lib = amd.load_library()
m = amd.GNS(20, 10, K, 0.9, True).cuda(); m.topology_check = 'first'
bu, li, ge = amd.synth.synth_grids(case, bt, seed=1, device='cuda')
ref = None
for mapping, pack in [(1, 0), (2, 1), (2, 2), (2, 4), (2, 5)]:
    amd.set_option('fwd_mapping', mapping); amd.set_option('gw_pack', pack)
    try:
        with torch.no_grad():
            for it in range(3): out = m(bu, li, ge)
            lib.gns_profile_enable(64)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for it in range(30): m(bu, li, ge)
            torch.cuda.synchronize(); t2 = time.perf_counter()
    except Exception as ex:
        print(f'mapping {mapping} pack {pack}: {ex}', flush=True)
        continue
    a, n = ctypes.c_float(), ctypes.c_int()
    lib.gns_profile_read(0, ctypes.byref(a), ctypes.byref(n))
    lib.gns_profile_enable(0)
    if ref is None:
        ref = [o.double() for o in out]
    err = max(float((o.double() - r).abs().max() / r.abs().max()) for o, r in zip(out, ref))
    print(f"case{case} x {bt} K={K} mapping {mapping} pack {pack}: fwd(eval) kernel {a.value / max(n.value, 1):.3f} ms   "
          f"call loop {(t2 - t0) / 30 * 1e3:.3f} ms   max rel diff vs lane mapping {err:.2e}", flush=True)
