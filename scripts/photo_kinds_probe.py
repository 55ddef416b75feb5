"""GPU box: where does real picture content cost time?  For each of N photo frames (photos.py) of one configuration: fps of the device loop
on THAT frame alone (4 contexts in flight, frames resident in HBM), share of uncertified pixels, share of tiles with a worklist and of
tiles whose list overflowed into the all-exact stage.  Usage: photo_kinds_probe.py [config=C2] [n=16] [frames per measurement=192]
-> gpurun_out/photo_kinds_<config>.json (+ a table on stdout)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-super-resolution-library_amd")]
import torch
import bench, photos, raisr_hip as R, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16
NF = int(sys.argv[3]) if len(sys.argv) > 3 else 192
wl = bench.Workload(cfg)
names = photos.available()
blobs = []
for p in range(wl.passes):
    bank, qstr, qcoh, qa = R.read_model_folder(wl.folder, wl.bits, p + 1)
    hb = R.pack_model_blob(bank, qstr, qcoh, qa)
    blobs.append(torch.from_numpy(hb).cuda())


def fence():
    torch.cuda.synchronize()


def measure(frame):
    dt, _, lanes, _, _ = bench.device_loop(R, torch, wl, 0, blobs, 4, [frame], NF, 1, 1, fence, False)
    for d in lanes:
        d.close()
    d = R.RaisrDevice(0, hooks=True)
    for p in range(wl.passes):
        d.set_model_blob_device(p, blobs[p].data_ptr(), blobs[p].numel())
    d.configure(wl.in_w, wl.in_h, wl.out_w, wl.out_h, bits=wl.bits, passes=wl.passes, mode=wl.mode, hash_variant=wl.asm)
    d.certify_debug(True, False)
    out = np.zeros((wl.out_h, wl.out_w), frame.dtype)
    d.process_host(frame, out)
    st = d.certify_stats()
    d.close()
    return NF / dt, st


rows = []
ref_fps, st = measure(synth.natural_y(wl.in_w, wl.in_h, wl.bits, seed=12345))
rows.append(("synthetic natural", "-", ref_fps, st))
for k in range(N):
    i = 13 * k % (len(names) * len(photos.VARIANTS))         # 13 is coprime with sources x variants
    fps, st = measure(photos.photo_y(wl.in_w, wl.in_h, wl.bits, i))
    rows.append((names[i % len(names)], photos.VARIANTS[(i // len(names)) % len(photos.VARIANTS)], fps, st))
res = []
print(f"{cfg}: {'source':18s} {'variant':10s} {'fps':>8s} {'vs synth':>8s} {'uncert %':>8s} {'listed %':>8s} {'overflow %':>10s} {'flat %':>7s}")
for nm, var, fps, st in rows:
    t = max(1, st["tiles"])
    r = {"source": nm, "variant": var, "fps": round(fps, 1), "rel": round(fps / ref_fps, 3), "uncertified_frac": round(st["uncertain"] / max(1, st["pixels"]), 5),
         "tiles_listed_frac": round(st["tiles_listed"] / t, 4), "tiles_overflow_frac": round(st["tiles_overflow"] / t, 4), "tiles_flat_frac": round(st["tiles_flat"] / t, 4)}
    res.append(r)
    print(f"    {nm:18s} {var:10s} {fps:8.0f} {r['rel']:8.3f} {100 * r['uncertified_frac']:8.2f} {100 * r['tiles_listed_frac']:8.1f} {100 * r['tiles_overflow_frac']:10.2f} {100 * r['tiles_flat_frac']:7.1f}")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"config": cfg, "frames_per_measurement": NF, "rows": res}, open(os.path.join(ROOT, "gpurun_out", f"photo_kinds_{cfg}.json"), "w"), indent=1)
