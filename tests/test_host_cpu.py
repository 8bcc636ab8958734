"""CPU (-m "not gpu"): host-side logic of the product package against the oracle, the C-ABI
library loads and exports every symbol include/kdip.h declares, no compute without a GPU, and
the N>1 sharding/gather path under gloo (world_size 2)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """include/kdip.h = the drop-in boundary, include/kdip_internal.h = test / diagnostic hooks: every declaration of
    both is exported by the library and bound in _lib.SIGNATURES, nothing else is bound, and the public header holds no hook."""
    import kdip_amd._lib as L
    lib = L.load()
    decl = {}
    for h in ("kdip.h", "kdip_internal.h"):
        hdr = open(os.path.join(ROOT, "include", h)).read()
        decl[h] = set(re.findall(r"\b(kdip_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)))
        assert decl[h], "no declarations parsed in " + h
    assert not [n for n in decl["kdip.h"] if n.startswith(("kdip_test_", "kdip_debug_"))]
    assert all(n.startswith(("kdip_test_", "kdip_debug_")) for n in decl["kdip_internal.h"]), decl["kdip_internal.h"]
    declared = decl["kdip.h"] | decl["kdip_internal.h"]
    assert declared == set(L.SIGNATURES), (declared ^ set(L.SIGNATURES))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.kdip_version() >= 100
    assert lib.kdip_profile_num_classes() > 0


def test_no_cpu_fallback():
    """Without a GPU every compute entry point of the product fails loudly."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import kdip_amd._lib as L
    import kdip_amd.unet as ku
    import kdip_amd.measurements as km
    import kdip_amd.sampling as ks
    with pytest.raises(L.KdipError):
        ku.UNetModel(image_size=64, model_channels=32, attention_resolutions="32", channel_mult=(1, 2))
    with pytest.raises(L.KdipError):
        km.get_operator("gaussian_blur", device="cpu", in_shape=(1, 3, 64, 64), kernel_size=61, intensity=3.0, sigma_s=0.05)
    with pytest.raises(L.KdipError):
        ks.sample_euler(lambda x, s: x, torch.zeros(1, 3, 8, 8), ks.get_sigmas_karras(4, 0.01, 80))


def _oracle_uses(path, skip_functions=()):
    """Every way a source file could reach the checker, found on its syntax tree (prose in comments / docstrings does not count):
    `import oracle[.x]`, `from oracle[.x] import ...`, relative imports that resolve to it, `__import__` / `importlib.import_module`
    of it, and any string literal naming an oracle path that is an ARGUMENT of a call (ctypes.CDLL, subprocess.*, open, os.path.join,
    sys.path.insert ...).  Functions named in skip_functions are pruned (bench.py's cpu_baseline leg is allowed to call the checker)."""
    import ast
    tree = ast.parse(open(path).read(), filename=path)
    hits = []
    docstrings = set()
    for node in ast.walk(tree):
        if isinstance(node, (ast.Module, ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)) and node.body and \
                isinstance(node.body[0], ast.Expr) and isinstance(node.body[0].value, ast.Constant) and isinstance(node.body[0].value.value, str):
            docstrings.add(id(node.body[0].value))

    def names_oracle(mod):
        return mod is not None and (mod == "oracle" or mod.startswith("oracle.") or ".oracle." in mod or mod.endswith(".oracle"))

    def visit(node):
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef)) and node.name in skip_functions:
            return
        if isinstance(node, ast.Import):
            hits.extend((node.lineno, "import " + a.name) for a in node.names if names_oracle(a.name))
        elif isinstance(node, ast.ImportFrom):
            if names_oracle(node.module) or any(a.name == "oracle" for a in node.names):
                hits.append((node.lineno, "from %s import ..." % node.module))
        elif isinstance(node, ast.Call):
            for a in list(node.args) + [k.value for k in node.keywords]:
                for c in ast.walk(a):
                    if isinstance(c, ast.Constant) and isinstance(c.value, str) and id(c) not in docstrings and \
                            re.search(r"(^|[/\\.\s\"'])oracle([/\\.]|$)", c.value):
                        hits.append((node.lineno, "call argument %r" % c.value))
        for ch in ast.iter_child_nodes(node):
            visit(ch)

    visit(tree)
    return hits


def test_product_never_imports_oracle():
    """The product path (the package, the harness, bench.py outside its cpu_baseline leg) never imports, loads, opens or executes
    anything under oracle/ -- checked on the syntax tree, so documentation may mention the checker."""
    pkg = os.path.join(ROOT, "k-diffusion-inverse-problems_amd")
    files = [os.path.join(pkg, fn) for fn in sorted(os.listdir(pkg)) if fn.endswith(".py")]
    files += [os.path.join(ROOT, "sample_condition.py"), os.path.join(ROOT, "kdip_amd.py")]
    for path in files:
        assert _oracle_uses(path) == [], (path, _oracle_uses(path))
    assert _oracle_uses(os.path.join(ROOT, "bench.py"), skip_functions=("cpu_baseline",)) == []
    # ... and the walker does see what it is meant to see
    hits = _oracle_uses(os.path.join(ROOT, "bench.py"))
    assert any("oracle" in h[1] for h in hits), "the guard no longer detects bench.py's cpu_baseline import"
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write('"""mentions oracle/make_golden.py in prose"""\nimport ctypes, subprocess\n# oracle in a comment\n'
                'def a():\n    ctypes.CDLL("oracle/_ref/libref.so")\n'
                'def b():\n    subprocess.run(["python", "-m", "oracle.make_golden"])\n'
                'def c():\n    import importlib; importlib.import_module("oracle.unet")\n')
    try:
        got = _oracle_uses(f.name)
        assert len(got) == 3, got
    finally:
        os.unlink(f.name)


def test_schedule_and_tables_match_oracle(gold):
    import kdip_amd.sampling as ks
    import kdip_amd.external as ke
    import kdip_amd.unet as ku
    from oracle import tables as otab
    g = gold("tables")
    assert torch.equal(ks.get_sigmas_karras(100, 0.01, 80), torch.from_numpy(g["sigmas100"]))
    D, O = ku.GaussianDiffusionTables(), otab.DiffusionTables()
    for nm in ("alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1",
               "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "log_betas"):
        assert np.array_equal(getattr(D, nm), getattr(O, nm)), nm
    den = ke.OpenAIDenoiser(None, D)
    probe = torch.from_numpy(g["probe"])
    assert torch.equal(den.sigma_to_t(probe), torch.from_numpy(g["t_frac"]))
    assert torch.equal(den.sigma_to_t(probe).long(), torch.from_numpy(g["t_floor"]))
    s = torch.tensor([0.3, 0.3]); s._kdip_host_value = 0.3
    assert torch.equal(den.sigma_to_t(s), O.sigma_to_t(torch.tensor([0.3, 0.3])))
    c_out, c_in = den.get_scalings(torch.tensor([2.0]))
    assert float(c_out) == -2.0 and abs(float(c_in) - 5 ** -0.5) < 1e-7


def test_synthetic_weights_and_plan_match_oracle():
    import kdip_amd.unet as ku
    from oracle import unet as ou
    a = ku.synthetic_state_dict(seed=0, out_cov=True, image_size=64, model_channels=32, num_res_blocks=1,
                                attention_resolutions="32", channel_mult=(1, 2))
    b = ou.init_state_dict(ou.UNetConfig(**ou.TINY), seed=0, out_cov=True)
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
    for cfg_p, cfg_o in ((ku.FFHQ_CONFIG, ou.FFHQ), (ku.IMAGENET_CONFIG, ou.IMAGENET)):
        sp, zp = ku.state_dict_shapes(**cfg_p)
        so, zo = ou.param_shapes(ou.UNetConfig(**cfg_o))
        assert sp == so and zp == zo
    n = sum(int(np.prod(s)) for s in ku.state_dict_shapes(**ku.FFHQ_CONFIG)[0].values())
    assert n == 93563910


def test_mask_generator_bit_exact(gold):
    import kdip_amd.measurements as km
    g = gold("operators")
    np.random.seed(0)
    m = km.MaskGenerator(mask_type="random", mask_prob_range=(0.5, 0.5), image_size=256)(torch.empty(1, 3, 256, 256))
    bits = np.unpackbits(g["inpainting.mask256_bits"]).reshape(256, 256)
    assert np.array_equal(m[0, 0].numpy().astype(np.uint8), bits)
    assert torch.equal(m[0, 0], m[0, 1]) and torch.equal(m[0, 0], m[0, 2])
    box = km.MaskGenerator(mask_type="box", mask_len_range=(128, 129))(torch.empty(1, 3, 256, 256))     # centred 128 x 128 box
    assert float(box.sum()) == 3 * (256 * 256 - 128 * 128) and float(box[0, 0, 64:192, 64:192].sum()) == 0
    with pytest.raises(NotImplementedError):
        km.MaskGenerator(mask_type="both", mask_len_range=(128, 129), mask_prob_range=(0.5, 0.5))(torch.empty(1, 3, 256, 256))


def test_resize_tables_match_oracle():
    import kdip_amd.measurements as km
    from oracle import operators as oops
    for n_in, n_out in ((256, 64), (64, 16)):
        w, f = km.cubic_resize_tables(n_in, n_out, 0.25)
        wo, fo = oops.resizer_contributions(n_in, n_out, 0.25)
        assert w.shape == wo.shape == (n_out, 16)
        assert np.array_equal(f, fo.astype(np.int32))
        assert np.array_equal(w, wo.astype(np.float32))


def test_transposed_resize_tables_are_the_adjoint():
    """The Resizer adjoint runs as a gather on transposed tables (kdip_amd.measurements.transpose_resize_tables): as dense matrices,
    transposed table == (forward table)^T exactly -- mirrored borders (an input index hit twice by one output) included."""
    from kdip_amd.measurements import cubic_resize_tables, transpose_resize_tables
    for n_in, n_out, scale in ((256, 64, 0.25), (64, 16, 0.25), (96, 32, 1.0 / 3), (64, 32, 0.5)):
        w, f = cubic_resize_tables(n_in, n_out, scale)
        A = np.zeros((n_out, n_in), np.float64)
        for o in range(n_out):
            for t in range(w.shape[1]):
                A[o, f[o, t]] += w[o, t]
        wt, ft = transpose_resize_tables(w, f, n_in)
        At = np.zeros((n_in, n_out), np.float64)
        for s_ in range(n_in):
            for j in range(wt.shape[1]):
                At[s_, ft[s_, j]] += wt[s_, j]
        assert wt.dtype == np.float32 and ft.dtype == np.int32 and ft.min() >= 0 and ft.max() < n_out
        assert np.array_equal(At, A.T)


def test_registries():
    import kdip_amd.measurements as km
    import kdip_amd.condition as kc
    import kdip_amd.transforms as kt
    assert set(km.__OPERATOR__) == {"motion_blur", "gaussian_blur", "super_resolution", "inpainting"}
    assert set(kc.__MAT_SOLVER__) == set(km.__OPERATOR__)
    assert set(kt.__OT__) == {"dct", "dwt"}
    with pytest.raises(NameError):
        km.get_operator("nope")
    with pytest.raises(NameError):
        km.register_operator("inpainting")(type("X", (), {}))
    f = kc.register_mat_solver("custom_op")(lambda *a, **k: None)
    assert kc.__MAT_SOLVER__.pop("custom_op") is f


def test_shard_range_and_compute_features_single():
    from kdip_amd.evaluation import shard_range, compute_features, DistEnv, psnr
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert shard_range(128, 7, 8) == (112, 128)
    env = DistEnv(init=False)
    env.world_size = 1
    calls = []

    def sample_fn(n):
        calls.append(n)
        return torch.full((n, 3, 4, 4), float(len(calls)))
    out = compute_features(env, sample_fn, lambda x: x, 5, 2)
    assert out.shape[0] == 5 and calls == [2, 2, 1]
    a = torch.zeros(1, 3, 4, 4); b = torch.full((1, 3, 4, 4), 0.2)
    assert abs(float(psnr(a, b)) - 20.0) < 1e-4


WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from kdip_amd.evaluation import DistEnv, compute_features, shard_range
env = DistEnv(backend="gloo")
assert env.world_size == 2
lo, hi = shard_range(6, env.rank, env.world_size)
def sample_fn(n):
    return torch.arange(lo, lo + n, dtype=torch.float32).view(n, 1, 1, 1).expand(n, 3, 2, 2).contiguous()
out = compute_features(env, sample_fn, lambda x: x, 6, 3)
assert out.shape == (6, 3, 2, 2), out.shape
assert out[:, 0, 0, 0].tolist() == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0], out[:, 0, 0, 0]
t = env.max_over_ranks(float(env.rank + 1))
assert t == 2.0
topo = env.topology()
assert topo["ranks"] == 2 and topo["distinct_devices"] == 2 and topo["backend"] == "gloo" and len(topo["devices"]) == 2, topo
env.barrier()
print("rank", env.rank, "ok")
'''


def test_compute_features_gloo_world2(tmp_path):
    """N>1 path: contiguous shards, no collective until the final all_gather (gloo on CPU)."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", str(script), ROOT]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


def test_metrics_psnr_ssim_against_scipy_restatement():
    """kdip_amd.metrics (compute_metrics of the caller harness, sample_condition_openai.py:41-49): SSIM against an
    independent scipy.ndimage restatement of skimage's algorithm (uniform 7x7 window, sample covariance, cropped mean,
    channel average); PSNR against its definition; averaging helper."""
    import numpy as np
    from scipy.ndimage import uniform_filter
    import kdip_amd.metrics as M
    g = torch.Generator().manual_seed(4)
    a = torch.rand(3, 40, 52, generator=g)
    b = (a + 0.1 * torch.randn(3, 40, 52, generator=g)).clip(0, 1)

    def ssim_ref(x, y, R=1.0, win=7):
        vals = []
        for c in range(x.shape[0]):
            X, Y = x[c].astype(np.float64), y[c].astype(np.float64)
            NP = win * win; cov = NP / (NP - 1.0)
            ux, uy = uniform_filter(X, win), uniform_filter(Y, win)
            vx = cov * (uniform_filter(X * X, win) - ux * ux)
            vy = cov * (uniform_filter(Y * Y, win) - uy * uy)
            vxy = cov * (uniform_filter(X * Y, win) - ux * uy)
            C1, C2 = (0.01 * R) ** 2, (0.03 * R) ** 2
            S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
            p = (win - 1) // 2
            vals.append(S[p:-p, p:-p].mean())
        return float(np.mean(vals))

    assert abs(M.structural_similarity(a, b) - ssim_ref(a.numpy(), b.numpy())) < 1e-9
    assert abs(M.structural_similarity(a, a) - 1.0) < 1e-12
    mse = float(((a.double() - b.double()) ** 2).mean())
    assert abs(M.peak_signal_noise_ratio(a, b) - 10 * np.log10(1 / mse)) < 1e-9
    x0 = (a * 2 - 1)[None]; hx = (b * 2 - 1)[None]
    m = M.compute_metrics(hx, x0)
    assert set(m) == {"psnr", "ssim"} and abs(m["psnr"] - M.peak_signal_noise_ratio(a, b)) < 1e-4
    avg = M.calculate_average_metric([{"psnr": 1.0, "ssim": 0.5}, {"psnr": 3.0}])
    assert avg == {"psnr": 2.0, "ssim": 0.5}


def test_normalize_state_dict_layouts():
    """plain OpenAI state_dict passes through; a Lightning-style payload (train_openai.py:86-87 key layout) is reduced to the
    EMA UNet + out_cov keys the loader expects."""
    import kdip_amd.unet as ku
    w = torch.zeros(2)
    plain = {"input_blocks.0.0.weight": w, "out.2.bias": w}
    assert ku.normalize_state_dict(plain).keys() == plain.keys()
    assert ku.normalize_state_dict({"state_dict": plain}).keys() == plain.keys()
    lit = {"state_dict": {"model.inner_model.out.2.bias": w + 1, "model_ema.inner_model.out.2.bias": w + 2, "model_ema.out_cov.weight": w + 3,
                          "model_ema.sigmas": w, "model.out_cov.weight": w, "ema_decay": w}}
    got = ku.normalize_state_dict(lit)
    assert set(got) == {"out.2.bias", "out_cov.weight"} and float(got["out.2.bias"][0]) == 2.0 and float(got["out_cov.weight"][0]) == 3.0
    got = ku.normalize_state_dict(lit, prefer_ema=False)
    assert float(got["out.2.bias"][0]) == 1.0


def test_harness_config_loaders(tmp_path):
    """sample_condition.py reads `file#entry` (configs/tasks.yaml, configs/models.json) and the reference's one-config-per-file
    layout, including `!!python/tuple` tags."""
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "sample_condition.py")).read()
    ns = {}
    exec(compile(src[src.index("def load_yaml"):src.index("def folder_of_images")], "loaders", "exec"), {"yaml": __import__("yaml"), "json": __import__("json")}, ns)
    t = ns["load_yaml"](os.path.join(root, "configs", "tasks.yaml#gaussian_deblur"))
    assert t == {"name": "gaussian_blur", "in_shape": [1, 3, 256, 256], "kernel_size": 61, "intensity": 3.0, "sigma_s": 0.05}
    assert set(ns["load_yaml"](os.path.join(root, "configs", "tasks.yaml"))) == {"gaussian_deblur", "motion_deblur", "super_resolution_4x", "inpainting", "inpainting_box", "gaussian_deblur_64", "inpainting_64"}
    m = ns["load_json"](os.path.join(root, "configs", "models.json#imagenet"))
    assert m["model"]["openai"] == {"num_channels": 256, "num_res_blocks": 2, "attention_resolutions": "8,16,32"}
    f = tmp_path / "one.yaml"
    f.write_text("name: super_resolution\nin_shape: !!python/tuple [1, 3, 256, 256]\nscale_factor: 4\nsigma_s: 0.05\n")
    one = ns["load_yaml"](str(f))
    assert one["in_shape"] == (1, 3, 256, 256) and one["scale_factor"] == 4


# ------------------------------------------------------------------ dataset / checkpoint IO ----
def test_folder_of_images_roundtrip(tmp_path):
    """K.utils.FolderOfImages + ToTensor + (x*2-1) (k_diffusion/utils.py:274-297, sample_condition_openai.py:138-143): sorted
    recursive listing, RGB conversion, [-1, 1] range; to_pil_image is its inverse up to 8-bit rounding."""
    import numpy as np
    from PIL import Image
    import sample_condition as sc
    rng = np.random.RandomState(0)
    names = ["b/2.png", "a/1.png", "c.jpg", "a/0.bmp", "notes.txt"]
    imgs = {}
    for n in names:
        p = tmp_path / n
        p.parent.mkdir(parents=True, exist_ok=True)
        if n.endswith(".txt"):
            p.write_text("not an image")
            continue
        a = rng.randint(0, 256, size=(16, 16, 3), dtype=np.uint8)
        Image.fromarray(a).save(p, quality=100) if n.endswith(".jpg") else Image.fromarray(a).save(p)
        imgs[str(p)] = a
    out = list(sc.folder_of_images(str(tmp_path)))
    order = sorted(imgs)
    assert len(out) == 4
    for t, path in zip(out, order):
        assert t.shape == (3, 16, 16) and t.dtype == torch.float32 and float(t.min()) >= -1 and float(t.max()) <= 1
        if not path.endswith(".jpg"):                     # lossless formats round-trip exactly
            back = np.asarray(sc.to_pil_image(t))
            assert np.array_equal(back, imgs[path])
    # grayscale / palette files are converted to RGB
    Image.fromarray(rng.randint(0, 256, size=(8, 8), dtype=np.uint8), mode="L").save(tmp_path / "gray.png")
    g = [t for t in sc.folder_of_images(str(tmp_path))]
    assert all(t.shape[0] == 3 for t in g)


def test_checkpoint_layouts_on_disk(tmp_path):
    """The two checkpoint layouts the reference's scripts load (sample_condition_openai.py:130-132: a plain state_dict .pt;
    sample_condition_openai_v2.py:116 / train_openai.py:86-87: a Lightning .ckpt whose 'state_dict' holds model_ema.inner_model.* and
    model_ema.out_cov.* next to optimizer-side keys) written to disk, read back with torch.load and normalised to the key layout
    kdip_unet_load expects."""
    import kdip_amd.unet as ku
    cfg = dict(image_size=64, model_channels=32, num_res_blocks=1, attention_resolutions="32", channel_mult=(1, 2))
    sd = ku.synthetic_state_dict(seed=3, out_cov=True, **cfg)
    plain = {k: v for k, v in sd.items() if not k.startswith("out_cov.")}
    torch.save(plain, tmp_path / "diffusion.pt")
    got = ku.normalize_state_dict(torch.load(tmp_path / "diffusion.pt", map_location="cpu"))
    assert set(got) == set(plain) and all(torch.equal(got[k], plain[k]) for k in plain)
    lightning = {"epoch": 3, "global_step": 1000, "pytorch-lightning_version": "2.0.0",
                 "state_dict": {**{"model.inner_model." + k: v + 1 for k, v in plain.items()},                      # the non-EMA copy must be ignored
                                **{"model_ema.inner_model." + k: v for k, v in plain.items()},
                                **{"model_ema.out_cov." + k[len("out_cov."):]: v for k, v in sd.items() if k.startswith("out_cov.")},
                                "model_ema.sigma_data": torch.tensor(0.5)},
                 "optimizer_states": [{}]}
    torch.save(lightning, tmp_path / "ffhq_dwt.ckpt")
    got = ku.normalize_state_dict(torch.load(tmp_path / "ffhq_dwt.ckpt", map_location="cpu"))
    assert set(got) == set(sd), sorted(set(got) ^ set(sd))[:5]
    assert all(torch.equal(got[k], sd[k]) for k in sd)
    shapes, _ = ku.state_dict_shapes(out_cov=True, **cfg)
    assert all(tuple(got[k].shape) == tuple(shapes[k]) for k in shapes)


def test_dtype_names_match_the_header():
    """kdip_amd._lib.DTYPES <-> include/kdip.h (KDIP_F32 / KDIP_BF16 / KDIP_BF16X3)."""
    import re
    import kdip_amd._lib as L
    hdr = open(os.path.join(ROOT, "include", "kdip.h")).read()
    for name, key in (("KDIP_F32", "f32"), ("KDIP_BF16", "bf16"), ("KDIP_BF16X3", "bf16x3")):
        m = re.search(name + r"\s*=\s*(\d+)", hdr)
        assert m and int(m.group(1)) == L.DTYPES[key], name
