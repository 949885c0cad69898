"""Ad-hoc kernel timing through the library's own HIP-event hooks: fwd/bwd kernel ms and the whole-step time for
case118 x 16384 under each training mapping.  usage: python tools/gpu_time.py [case] [batch] [K] [mappings e.g. 1,2:1,2:4]"""
import sys, os, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import opf_graph_neural_solver_amd as amd
lib = amd.load_library()
case = int(sys.argv[1]) if len(sys.argv) > 1 else 118
bt = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
K = int(sys.argv[3]) if len(sys.argv) > 3 else 4
combos = sys.argv[4] if len(sys.argv) > 4 else '1,2:1,2:4'
m = amd.GNS(20, 10, K, 0.9, True).cuda(); m.topology_check = 'first'
bu, li, ge = amd.synth.synth_grids(case, bt, seed=1, device='cuda')
for c in combos.split(','):
    mapping, pack, variant = (int(x) for x in (c.split(':') + ['0', '2'][len(c.split(':')) - 1:])[:3])
    amd.set_option('train_mapping', mapping); amd.set_option('gw_pack', pack); amd.set_option('bwd_variant', variant)
    for it in range(2):
        out = m(bu, li, ge); out[2].mean().backward(); m.zero_grad()
    lib.gns_profile_enable(16)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in range(10):
        out = m(bu, li, ge); out[2].mean().backward(); m.zero_grad()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    a, n = ctypes.c_float(), ctypes.c_int()
    lib.gns_profile_read(0, ctypes.byref(a), ctypes.byref(n)); f = a.value / max(n.value, 1)
    lib.gns_profile_read(1, ctypes.byref(a), ctypes.byref(n)); b = a.value / max(n.value, 1)
    lib.gns_profile_enable(0)
    print(f"case{case} x {bt} K={K} train_mapping {mapping} pack {pack} bwd_variant {variant}: fwd(train) {f:.3f} ms   bwd {b:.3f} ms   fwd+bwd loop {(t1 - t0) / 10 * 1e3:.3f} ms", flush=True)
