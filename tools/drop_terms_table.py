"""One line of profiles/r06/precise_attention_planes.txt: for the library in VITX_LIB (a build of attention_stream.hip with -DAP_DROP=n), the F16 parity
mode's (a) attention kernel on f32 inputs that are not fp16-representable against the oracle (the numbers test_attention_precise_on_non_representable_inputs
asserts), (b) ViT-B/16 batch 256 on the x4-head fixture and (c) on the bench's x8-head weights, max |dp| against the reference semantics on 24 / 48 rows with
the oracle's own summation-order noise, (d) ms per forward and the attention kernel's microseconds per launch.  Whole graph (last_layer_all_rows = 1)."""
import dataclasses, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import _pkg
pkg = _pkg.load()
from vitcpp_amd import binding as B
from oracle import oracle as O

tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(B.LIB_PATH)
out = {"variant": tag}
# (a) the attention op alone
n_img, N, H = 2, 197, 4; D = H * 64
rng = np.random.default_rng(42)
qkv32 = (rng.standard_normal((n_img * N, 3 * D)) * 0.8).astype(np.float32)
ref = O.attention(qkv32, n_img, N, D, H, O.REF)
dq = torch.from_numpy(qkv32).cuda()
d_hi = torch.from_numpy(qkv32.astype(np.float16)).cuda()
d_out = torch.zeros((n_img * N, D), dtype=torch.float16, device="cuda")
B.check(B.lib().vitx_op_attention_f32(dq.data_ptr(), d_out.data_ptr(), n_img, N, D, H, None), "vitx_op_attention_f32")      # splits into hi / lo planes, then the parity mode's kernel
torch.cuda.synchronize()
d = np.abs(d_out.float().cpu().numpy() - ref)
out16 = torch.zeros_like(d_out)
B.check(B.lib().vitx_op_attention(B.F16, d_hi.data_ptr(), out16.data_ptr(), n_img, N, D, H, None))
torch.cuda.synchronize()
d16 = np.abs(out16.float().cpu().numpy() - ref)
out["attention_op"] = {"max": float(d.max()), "mean": float(d.mean()), "rounded_operands_mean": float(d16.mean()), "test_would_pass": bool(d.max() <= 3e-3 and d.mean() <= 3e-4 and d.mean() < 0.75 * d16.mean())}

# (b), (c) whole forwards
def forward_case(head_scale, n_rows, seed_imgs):
    name = "vit_base_patch16_224"
    path = pkg.synth.cached_synthetic(name, head_scale=head_scale)
    if seed_imgs is None:      # bench.py's batch
        g = torch.Generator(device="cpu").manual_seed(4321)
        u8 = torch.randint(0, 256, (256, 224, 224, 3), generator=g, dtype=torch.uint8)
        imgs = ((u8.float() - torch.tensor(pkg.synth.IMAGENET_MEAN)) / torch.tensor(pkg.synth.IMAGENET_STD)).contiguous().numpy()
    else:
        imgs = pkg.synth.normalize_u8(pkg.synth.synthetic_images_u8(256, 224, seed=seed_imgs))
    m = B.Model(path); c = B.Context(m, device=0, max_batch=256, dtype=B.F16, last_layer_all_rows=1)
    rows = sorted(set(c.boundary_rows(256)) | set(int(i) for i in np.linspace(0, 255, n_rows).round()))[:max(n_rows, 6)]
    d_in = torch.from_numpy(imgs).cuda(); d_p = torch.empty((256, m.num_classes), device="cuda")
    st = torch.cuda.Stream()
    for _ in range(5): c.forward_device(d_in.data_ptr(), 256, d_p.data_ptr(), 0, st.cuda_stream)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): c.forward_device(d_in.data_ptr(), 256, d_p.data_ptr(), 0, st.cuda_stream)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
    c.profile_enable(True); c.forward_device(d_in.data_ptr(), 256, d_p.data_ptr(), 0, st.cuda_stream); torch.cuda.synchronize()
    pr = {p["name"]: p for p in c.profile_read()}; c.profile_enable(False)
    got = d_p.cpu().numpy()[rows]
    om = O.OracleModel(path)
    _, rp = om.forward(imgs[rows], O.REF)
    _, xp = om.forward(imgs[rows], dataclasses.replace(O.REF, dot_exact=1))
    om.close(); c.close(); m.close()
    a = pr.get("attention", {})
    return {"rows": len(rows), "max_dprob_vs_ref": float(np.abs(got - rp).max()), "noise_floor": float(np.abs(xp - rp).max()), "top1_equal": bool((got.argmax(1) == rp.argmax(1)).all()),
            "ms_per_forward": round(ms, 3), "attention_us_per_launch": round(a.get("total_ms", 0) / max(1, a.get("launches", 1)) * 1e3, 1)}
out["head_x4"] = forward_case(4.0, 24, 2025)
out["head_x8_bench_rows"] = forward_case(8.0, 48, None)
print("PLANES " + json.dumps(out), flush=True)
