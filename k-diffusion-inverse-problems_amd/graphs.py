"""hipGraph replay of guided-denoiser calls.

A production sampler runs the same sigma schedule for every batch, so the ~450 kernel launches of one guided call (UNet forward,
x0 epilogue, closed-form mat-solver, cotangent, hand-written UNet VJP, guidance combine) are captured ONCE per (sigma, batch
shape) into a hipGraph (torch.cuda.CUDAGraph drives hipStreamBeginCapture on the stream the C-ABI launches go to) and replayed
for every later batch: one graph launch instead of hundreds of kernel launches per call.

What is capture-safe and what is not:
  * every libkdip_hip entry point on the closed-form path is (no allocation after the first call at a batch size -- workspaces are
    planned per batch on the host --, no host synchronisation, hipMemsetAsync only);
  * the CG branch (tensor-valued covariance below `mle_sigma_thres`: 42 of the 199 calls of BASELINE configs[1]) reads its
    per-sample convergence flags back to the host every second iteration, so those calls stay eager.

Captured graphs hold raw pointers into the UNet handle's workspace arenas and to the denoiser's measurement tensors.  The handle
regrows (frees + re-allocates) its arenas when a call at a new batch shape does not fit, so every graph records the handle's
`workspace_generation()` and the measurement tensors' identities; when either changes all graphs are dropped and re-captured on
their next use (`invalidations` counts this).

`GraphedDenoiser(den)` is a drop-in for the ConditionDenoiser it wraps (`model(x, sigma)` of the samplers)."""
import torch

from .external import sigma_host


class GraphedDenoiser:
    def __init__(self, den, enabled=True):
        self.den = den
        self.enabled = enabled
        self._graphs = {}          # (sigma, shape) -> (graph, static_x, static_out)
        self.replays = 0
        self.eager_calls = 0
        self.invalidations = 0
        self._bound = None         # (workspace generation, measurement tensor identities) the graphs were captured against

    def __getattr__(self, name):   # plug-in surface of the wrapped denoiser (operator, guidance, ...)
        return getattr(self.den, name)

    def _bindings(self):
        """What a captured graph baked in besides its static input: the UNet handle's workspace arenas (raw device pointers) and
        the denoiser's measurement tensors."""
        den = self.den
        model = getattr(den, "inner_model", None)
        gen = model.workspace_generation() if hasattr(model, "workspace_generation") else 0
        meas = tuple((t.data_ptr(), tuple(t.shape)) for t in (getattr(den, n, None) for n in ("y", "y_flatten", "_y1", "_yf1"))
                     if isinstance(t, torch.Tensor))
        return (gen, meas)

    def _validate(self):
        """Drop every captured graph when the handle re-allocated its arenas (a call at a larger batch shape, eager or captured,
        regrows them and frees the old ones: the old graphs would read and write freed memory) or the measurement was rebound."""
        b = self._bindings()
        if b != self._bound:
            if self._graphs:
                self.invalidations += 1
            self._graphs.clear()
            self._bound = b

    def capturable(self, sigma_value):
        den = self.den
        if den.guidance in ("stsl", "stsl+mle"):                 # draws Hutchinson probes with the device generator per call
            return False
        tensor_cov = getattr(den, "x0_cov_type", None) in ("convert", "tmpd") or not hasattr(den, "x0_cov_type")   # V2: learned variances
        uses_solver = den.guidance in ("I", "II", "autoI", "dps+mle", "pgdm+mle")
        if uses_solver and tensor_cov and sigma_value < den.mle_sigma_thres:
            return False                                          # CG branch: host reads the convergence flags
        if getattr(den, "x0_cov_type", None) == "tmpd":
            return False
        return True

    def __call__(self, x, sigma):
        s = sigma_host(sigma)
        if not self.enabled or not self.capturable(s):
            self.eager_calls += 1
            return self.den(x, sigma)
        key = (s, tuple(x.shape), x.device.index)
        self._validate()
        ent = self._graphs.get(key)
        if ent is None:
            sx = x.detach().clone().contiguous()
            ssig = sx.new_full([sx.shape[0]], float(s))
            ssig._kdip_host_value = float(s)
            self.den(sx, ssig)                                    # eager warm-up: sizes workspaces, sets kernel attributes
            torch.cuda.current_stream().synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):     # (capture runs on torch's capture stream = current stream)
                so = self.den(sx, ssig)
            ent = (g, sx, so)
            if self._bindings() != self._bound:                   # the warm-up call itself regrew the arenas: older graphs are stale,
                self._graphs.clear()                              # this one was captured against the new arenas
                self._bound = self._bindings()
            self._graphs[key] = ent
        g, sx, so = ent
        sx.copy_(x)
        g.replay()
        self.replays += 1
        return so.clone()

    forward = __call__
