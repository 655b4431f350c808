#!/usr/bin/env python3
"""Throughput of the device svb-zd decoder on synthetic reads (bytes in: blob, bytes out: 2 B/sample)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1000)
    ap.add_argument("--read-len", type=int, default=100000)
    ap.add_argument("--replicate", type=int, default=20, help="replicate the encoded reads to enlarge the batch")
    a = ap.parse_args()
    import torch
    from sigtk_amd import api, blow5, device
    dev = torch.device("cuda", 0)
    L = api.load_library()
    reads, _, _, _ = api.synth_reads_host(a.reads, a.read_len, 9, 0)
    blobs = [blow5.svb_zd_encode(r) for r in reads] * a.replicate
    counts = [a.read_len] * len(blobs)
    out, st = device.svbzd_decode(blobs, counts, dev)
    torch.cuda.synchronize()
    assert int((st[:len(blobs)] != 0).sum().item()) == 0
    o = int(out.offsets_host[5]); assert np.array_equal(out.samples[o:o + a.read_len].cpu().numpy(), reads[5])
    # time the kernel alone on resident buffers
    n = len(blobs)
    blens = np.array([len(b) for b in blobs], dtype=np.uint32)
    boffs = np.zeros(n, dtype=np.int64); boffs[1:] = np.cumsum((blens[:-1].astype(np.int64) + 15) // 16 * 16)
    host = np.zeros(int(boffs[-1] + blens[-1]) + 16, dtype=np.uint8)
    for i, b in enumerate(blobs):
        host[int(boffs[i]):int(boffs[i]) + len(b)] = np.frombuffer(b, dtype=np.uint8)
    d_blobs = torch.from_numpy(host).to(dev); d_boffs = torch.from_numpy(boffs).to(dev)
    d_blens = torch.from_numpy(blens.astype(np.int32)).to(dev)
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    stream = int(torch.cuda.current_stream().cuda_stream)
    def run():
        api.check(L.sgk_svbzd_decode(d_blobs.data_ptr(), d_boffs.data_ptr(), d_blens.data_ptr(), n, out.samples.data_ptr(),
                                     out.offsets.data_ptr(), out.lengths.data_ptr(), status.data_ptr(), stream))
    run(); torch.cuda.synchronize()
    L.sgk_profile_reset(); L.sgk_profile_enable(1)
    for _ in range(5): run()
    torch.cuda.synchronize(); L.sgk_profile_enable(0)
    ms = {k: v[0] / v[1] for k, v in api.profile_read().items()}["k_svbzd_decode"]
    S = n * a.read_len
    byts = int(blens.astype(np.int64).sum()) + 2 * S
    print(json.dumps({"kernel": "k_svbzd_decode", "reads": n, "samples": S, "blob_bytes_per_sample": round(float(blens.sum()) / S, 3),
                      "ms": round(ms, 4), "samples_per_s": round(S / ms * 1e3, 1), "GBps": round(byts / ms / 1e6, 1),
                      "hbm_frac": round(byts / ms / 1e6 / 8000.0, 4)}))


    # ---- the encoder on the decoded samples (size pass + encode pass)
    blens_out = torch.zeros(n, dtype=torch.int32, device=dev)
    enc_blobs = torch.zeros(int(boffs[-1] + blens[-1]) + 64, dtype=torch.uint8, device=dev)

    def run_enc():
        api.check(L.sgk_svbzd_size(out.samples.data_ptr(), out.offsets.data_ptr(), out.lengths.data_ptr(), n,
                                   blens_out.data_ptr(), stream))
        api.check(L.sgk_svbzd_encode(out.samples.data_ptr(), out.offsets.data_ptr(), out.lengths.data_ptr(), n,
                                     enc_blobs.data_ptr(), d_boffs.data_ptr(), blens_out.data_ptr(), stream))
    run_enc(); torch.cuda.synchronize()
    assert torch.equal(blens_out, d_blens)
    o5 = int(boffs[5]); assert enc_blobs[o5:o5 + len(blobs[5])].cpu().numpy().tobytes() == blobs[5]
    L.sgk_profile_reset(); L.sgk_profile_enable(1)
    for _ in range(5): run_enc()
    torch.cuda.synchronize(); L.sgk_profile_enable(0)
    pr = {k: v[0] / v[1] for k, v in api.profile_read().items()}
    ms_e = pr["k_svbzd_size"] + pr["k_svbzd_encode"]
    print(json.dumps({"kernel": "k_svbzd_size + k_svbzd_encode", "reads": n, "samples": S, "ms": round(ms_e, 4),
                      "kernels_ms": {k: round(v, 4) for k, v in pr.items()}, "samples_per_s": round(S / ms_e * 1e3, 1),
                      "GBps": round(byts / ms_e / 1e6, 1), "hbm_frac": round(byts / ms_e / 1e6 / 8000.0, 4)}))


if __name__ == "__main__":
    main()
