#!/usr/bin/env python3
"""development: k_inflate on N zlib-compressed svb-zd records of 100 000-sample reads (what a BLOW5 batch holds):
ms per launch, inflated GB/s, samples/s.   python tools/bench_inflate.py [--reads 1280]"""
import argparse, json, os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sigtk_amd import api, blow5, device
from sigtk_amd.device import _ptr, _stream_ptr

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=1280)
ap.add_argument("--read-len", type=int, default=100000)
ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
L = api.load_library()
reads, *_ = api.synth_reads_host(64, a.read_len, 9, 0)
blobs = [blow5.svb_zd_encode(r) for r in reads]
streams = [zlib.compress(b"\x24\x00" + b"x" * 36 + bytes(44) + b) for b in blobs]
n = a.reads
sel = [streams[i % 64] for i in range(n)]
raw = [len(blobs[i % 64]) + 82 for i in range(n)]
in_off = np.zeros(n, dtype=np.uint64); in_len = np.asarray([len(s) for s in sel], dtype=np.uint32)
in_off[1:] = np.cumsum((in_len[:-1].astype(np.uint64) + 3) // 4 * 4)
blob = np.zeros(int(in_off[-1]) + int(in_len[-1]) + 8, dtype=np.uint8)
for r, s in enumerate(sel):
    blob[int(in_off[r]):int(in_off[r]) + len(s)] = np.frombuffer(s, dtype=np.uint8)
caps = np.asarray(raw, dtype=np.uint32)
out_off = np.zeros(n, dtype=np.uint64); out_off[1:] = np.cumsum((caps[:-1].astype(np.uint64) + 15) // 16 * 16)
dev = torch.device("cuda", 0)
t = lambda x, dt: torch.from_numpy(x.view(dt)).to(dev)
d_in, d_ioff, d_ilen = torch.from_numpy(blob).to(dev), t(in_off, np.int64), t(in_len, np.int32)
d_out = torch.zeros(int(out_off[-1]) + int(caps[-1]) + 16, dtype=torch.uint8, device=dev)
d_ooff, d_caps = t(out_off, np.int64), t(caps, np.int32)
d_olen = torch.zeros(n, dtype=torch.int32, device=dev); d_st = torch.zeros(n, dtype=torch.int32, device=dev)
def run():
    api.check(L.sgk_inflate(_ptr(d_in), _ptr(d_ioff), _ptr(d_ilen), n, _ptr(d_out), _ptr(d_ooff), _ptr(d_caps), _ptr(d_olen), _ptr(d_st), _stream_ptr()), "sgk_inflate")
run(); torch.cuda.synchronize()
assert int(d_st.abs().sum().item()) == 0
L.sgk_profile_enable(1)
for _ in range(a.steps): run()
torch.cuda.synchronize()
ms = api.profile_read()["k_inflate"]; ms = ms[0] / ms[1]
print(json.dumps({"reads": n, "compressed_mb": round(float(in_len.sum()) / 1e6, 1), "inflated_mb": round(float(caps.sum()) / 1e6, 1), "k_inflate_ms": round(ms, 3),
                  "inflated_GB_per_s": round(float(caps.sum()) / ms / 1e6, 2), "samples_per_s": round(n * a.read_len / ms * 1e3)}))
