"""DEV-ONLY helper: import the *reference* (/root/reference) in the build container.

TEST INFRASTRUCTURE. Used only by oracle/make_golden.py to (a) validate the CPU
restatement in oracle/ against the real reference and (b) emit tests/golden/*.npz.
It is never imported by the product, by tests, by bench.py or by smoke(): the
reference does not exist on the GPU box.

Accommodations: empty stand-ins for third-party modules that
are imported at module scope by the reference but never executed on the hot path,
and a rebinding of scipy's `cg(tol=...)` keyword (removed in scipy >= 1.14) to the
legacy criterion ||r|| <= tol*||b|| (`rtol=tol, atol=0`).
"""
import os
import sys
import types
import contextlib

REF = os.environ.get("KDIP_REFERENCE", "/root/reference")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []  # behave like a package so `import a.b` works
    sys.modules[name] = m
    return m


def install_shims():
    import torch
    import scipy.io
    import scipy.sparse.linalg as spla

    tv = _stub("torchvision", torch=torch)
    _stub("torchvision.utils", make_grid=None)
    _stub("torchvision.transforms")
    _stub("torchvision.transforms.functional")
    tv.utils = sys.modules["torchvision.utils"]
    tv.transforms = sys.modules["torchvision.transforms"]
    tv.datasets = _stub("torchvision.datasets")
    # PyWavelets: the real package lives in this image's second interpreter (/opt/conda/bin/python3.9); its four functions the
    # reference calls (condition/utils.py:116-132) are forwarded there (oracle/pywt_bridge.py).  Without that interpreter: empty stub.
    from . import pywt_bridge
    if pywt_bridge.available():
        _stub("pywt", wavedec2=pywt_bridge.wavedec2, coeffs_to_array=pywt_bridge.coeffs_to_array,
              array_to_coeffs=pywt_bridge.array_to_coeffs, waverec2=pywt_bridge.waverec2, __kdip_bridge__=True)
    else:
        _stub("pywt")
    gp = _stub("gpytorch", LinearOperator=object)
    _stub("gpytorch.distributions", MultivariateNormal=object)
    _stub("hdf5storage", loadmat=scipy.io.loadmat)
    _stub("skimage"); _stub("skimage.transform"); _stub("skimage.metrics")

    def _merge(a, b):
        out = dict(a)
        for k, v in b.items():
            out[k] = _merge(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else v
        return out
    _stub("jsonmerge", merge=_merge)
    _stub("cleanfid"); _stub("cleanfid.inception_torchscript", InceptionV3W=object)
    _stub("clip"); _stub("resize_right", resize=None)
    _stub("torchdiffeq", odeint=None); _stub("torchsde")
    _stub("cv2"); _stub("blobfile"); _stub("mpi4py", MPI=None)
    _stub("lightning"); _stub("lpips")
    _stub("dctorch"); _stub("dctorch.functional")
    _stub("kornia"); _stub("wandb")


@contextlib.contextmanager
def reference_cwd():
    """The reference loads PSFs by cwd-relative paths (measurements.py:95,134,173)."""
    old = os.getcwd()
    os.chdir(REF)
    try:
        yield
    finally:
        os.chdir(old)


def import_reference():
    """Returns a namespace with the reference modules used on the hot path."""
    install_shims()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    with reference_cwd():
        import condition.condition as cc
        import condition.measurements as cm
        import condition.utils as cu
        import k_diffusion.sampling as ks
        import k_diffusion.external as ke
        import guided_diffusion.script_util as su
        import guided_diffusion.gaussian_diffusion as gd
        import condition.diffpir_utils.utils_model as um
        import condition.diffpir_utils.utils_sisr as sisr
    import scipy.sparse.linalg as spla

    def legacy_cg(A, b, tol=1e-5, maxiter=None):
        return spla.cg(A, b, rtol=tol, atol=0.0, maxiter=maxiter)
    cc.cg = legacy_cg
    ns = types.SimpleNamespace(cc=cc, cm=cm, cu=cu, ks=ks, ke=ke, su=su, gd=gd, um=um, sisr=sisr)
    return ns
