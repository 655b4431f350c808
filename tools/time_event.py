"""development: time sgk_event on a device-resident synthetic batch with the library SIGTK_AMD_LIB points to
(tools/build_variant.sh); prints k_event / fallback ms, the status counters and the diagnostic counters."""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sigtk_amd import api, device

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=10000)
ap.add_argument("--read-len", type=int, default=100000)
ap.add_argument("--rna", type=int, default=0)
ap.add_argument("--kind", type=int, default=None)
ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda", 0)
L = api.load_library()
b = device.synth_reads(a.reads, a.read_len, seed=1, kind=a.rna if a.kind is None else a.kind, device=dev)
arena = device.EventArena(b)
device.event(b, arena, a.rna)
torch.cuda.synchronize()
L.sgk_profile_enable(1)
for _ in range(a.steps):
    device.event(b, arena, a.rna)
torch.cuda.synchronize()
prof = api.profile_read()
st = arena.status()
res = {"lib": os.environ.get("SIGTK_AMD_LIB", "default"), "reads": a.reads, "read_len": a.read_len, "rna": a.rna,
       "ms": {k: round(v[0] / max(v[1], 1), 4) for k, v in prof.items()},
       "events": int(st.n_events_total), "fallback": int(st.n_fallback_reads), "rerun": int(st.n_rerun_passes),
       "replays": int(st.n_long_replays)}
if hasattr(L, "sgk_debug_fp_counters"):
    out = (C.c_ulonglong * 10)()
    L.sgk_debug_fp_counters(out, 0)
    res["fp_calls_per_step"] = out[0] / a.steps
    res["fp_exact_per_step"] = out[1] / a.steps
if hasattr(L, "sgk_debug_event_why"):
    why = (C.c_uint32 * 5)()
    L.sgk_debug_event_why.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.sgk_debug_event_why(C.c_void_p(arena.ws.data_ptr()), why, None)
    res["why"] = list(why)
print(json.dumps(res))
