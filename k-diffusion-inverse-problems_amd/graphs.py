"""hipGraph replay of guided-denoiser calls.

A production sampler runs the same sigma schedule for every batch, so the ~450 kernel launches of one guided call (UNet forward,
x0 epilogue, closed-form mat-solver, cotangent, hand-written UNet VJP, guidance combine) are captured ONCE per (sigma, batch
shape) into a hipGraph (torch.cuda.CUDAGraph drives hipStreamBeginCapture on the stream the C-ABI launches go to) and replayed
for every later batch: one graph launch instead of hundreds of kernel launches per call.

What is capture-safe and what is not:
  * every libkdip_hip entry point on the closed-form path is (no allocation after the first call at a batch size -- workspaces are
    planned per batch on the host --, no host synchronisation, hipMemsetAsync only);
  * the CG branch (tensor-valued covariance below `mle_sigma_thres`: 42 of the 199 calls of BASELINE configs[1]) reads its
    per-sample convergence flags back to the host every second iteration when run eagerly.  For capture the operator is put
    into FIXED-TRIP mode (kdip_op_set_cg_fixed_trips): the eager warm-up call reports the iterations this sigma needs, the graph
    runs 1.5 x that + 4 iterations with no host read -- samples freeze themselves on the device when they converge, so surplus
    iterations change nothing --, and a sticky device counter records replays whose last residual check still found an active
    sample.  With `cg_check="each"` (default) the wrapper reads that counter after every fixed-trip replay (one stream
    synchronisation per CG call instead of one every second iteration); a replay that was not converged is warned about, redone
    eagerly with the adaptive solver -- the caller never receives a truncated solve -- and its graph is re-captured with twice the
    trips.  `cg_check="deferred"` leaves the polling to the caller (`cg_unconverged()`, e.g. once per sampler run).

Captured graphs hold raw pointers into the UNet handle's workspace arenas, the operator's solver workspace, the fused call's
workspace / timestep vector and the denoiser's measurement tensors.  The handles regrow (free + re-allocate) their buffers when a
call at a new batch shape does not fit, so every graph records the UNet's and the operator's `workspace_generation()` and the
tensors' identities; when any of them changes all graphs are dropped and re-captured on their next use (`invalidations`).

`GraphedDenoiser(den)` is a drop-in for the ConditionDenoiser it wraps (`model(x, sigma)` of the samplers)."""
import torch

from .external import sigma_host


class GraphedDenoiser:
    def __init__(self, den, enabled=True, capture_cg=True, cg_margin=1.5, cg_check="each"):
        assert cg_check in ("each", "deferred")
        self.den = den
        self.enabled = enabled
        self.capture_cg = capture_cg   # capture the CG branch too, as fixed-trip solves (trips = margin * iterations of the warm-up + 4)
        self.cg_margin = cg_margin
        self.cg_check = cg_check       # "each": poll the unconverged counter after every fixed-trip replay, redo + re-capture on a miss
        self.cg_trips = {}
        self._cg_floor = {}            # key -> trips an earlier re-capture needed (kept across graph invalidations)
        self.cg_redone = 0             # fixed-trip replays that were not converged and were redone eagerly
        self.x3_redone = 0             # dtype "f16x3": replays whose fp16-window flag came up and were redone eagerly (bf16-headed)
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
        op = getattr(den, "operator", None)
        opgen = op.workspace_generation() if hasattr(op, "workspace_generation") else 0        # solver buffers (CG vectors, FFT scratch)
        meas = tuple((t.data_ptr(), tuple(t.shape)) for t in (getattr(den, n, None) for n in ("y", "y_flatten", "_y1", "_yf1", "_fused_ws", "_fused_t"))
                     if isinstance(t, torch.Tensor))                                                # + the fused call's workspace / timestep vector
        return (gen, opgen, meas)

    def _validate(self):
        """Drop every captured graph when the handle re-allocated its arenas (a call at a larger batch shape, eager or captured,
        regrows them and frees the old ones: the old graphs would read and write freed memory) or the measurement was rebound."""
        b = self._bindings()
        if b != self._bound:
            if self._graphs:
                self.invalidations += 1
            self._graphs.clear()
            self._bound = b

    def _is_cg_branch(self, sigma_value):
        den = self.den
        tensor_cov = getattr(den, "x0_cov_type", None) in ("convert", "tmpd") or not hasattr(den, "x0_cov_type")   # V2: learned variances
        uses_solver = den.guidance in ("I", "II", "autoI", "dps+mle", "pgdm+mle")
        return uses_solver and tensor_cov and sigma_value < den.mle_sigma_thres

    def capturable(self, sigma_value):
        den = self.den
        if den.guidance in ("stsl", "stsl+mle"):                 # draws Hutchinson probes with the device generator per call
            return False
        if getattr(den, "x0_cov_type", None) == "tmpd":
            return False
        if self._is_cg_branch(sigma_value):                       # CG branch: capturable in fixed-trip mode only
            return self.capture_cg and hasattr(den.operator, "set_cg_fixed_trips")
        return True

    def cg_unconverged(self):
        """Number of replayed / captured fixed-trip CG solves, since the last query, that stopped with an unconverged sample: check
        once per sampler run (one stream synchronisation) -- 0 means every replay was as converged as the adaptive solver."""
        op = getattr(self.den, "operator", None)
        return op.cg_unconverged() if hasattr(op, "cg_unconverged") else 0

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
            cg = self._is_cg_branch(s)
            if cg:      # the adaptive warm-up call told how many CG iterations this sigma needs: capture that many + a margin
                op = self.den.operator
                need = max([i for i in getattr(op, "cg_iters", [0]) if i >= 0] or [0])
                # (never fewer than an earlier re-capture of this sigma asked for, never more than the solver's own iteration cap)
                trips = min(max(int(need * self.cg_margin) + 4, self._cg_floor.get(key, 0)), self.CG_MAXITER)
                op.set_cg_fixed_trips(trips)
                self.cg_trips[key] = trips
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g, capture_error_mode="thread_local"):     # (capture runs on torch's capture stream = current stream)
                    so = self.den(sx, ssig)
            finally:
                if cg:
                    self.den.operator.set_cg_fixed_trips(0)       # eager calls keep the adaptive solver
            ent = (g, sx, so)
            if self._bindings() != self._bound:                   # the warm-up call itself regrew the arenas: older graphs are stale,
                if self._graphs:                                  # this one was captured against the new arenas
                    self.invalidations += 1
                self._graphs.clear()
                self._bound = self._bindings()
            self._graphs[key] = ent
        g, sx, so = ent
        sx.copy_(x)
        g.replay()
        self.replays += 1
        # dtype "f16x3": a captured call cannot poll its fp16-window flag (no host reads under capture), so the replay is polled here
        # -- one stream synchronisation, as for the fixed-trip CG counter -- and a flagged call is redone eagerly, where
        # UNetModel.guarded() falls back to the bf16-headed arithmetic
        model = getattr(self.den, "inner_model", None) or getattr(getattr(self.den, "denoiser", None), "inner_model", None)
        if getattr(model, "x3_guard", False) and getattr(model, "dtype", None) == "f16x3" and (model.x3_saturated() & 5):
            self.x3_redone += 1
            self.eager_calls += 1
            return self.den(x, sigma)
        if key in self.cg_trips and self.cg_check == "each" and self.cg_unconverged() > 0:
            # the fixed trip count taken from the warm-up input was too small for THIS input: the reference iterates to tolerance or
            # maxiter and warns (condition.py:343-347), so do the same -- adaptive solve now, more trips for the next replay
            from warnings import warn
            warn(f"fixed-trip CG graph at sigma={s:.4g} ({self.cg_trips[key]} iterations) left a sample unconverged: call redone eagerly, graph re-captured")
            out = self.den(x, sigma)
            self.cg_redone += 1
            self.eager_calls += 1
            self._recapture_cg(key, self.cg_trips[key] * 2)
            return out
        return so.clone()

    CG_MAXITER = 1000                  # the mat-solvers' maxiter (condition/condition.py:343,379,432)

    def _recapture_cg(self, key, trips):
        g_old, sx, so_old = self._graphs.pop(key)
        trips = min(max(trips, self.cg_trips.get(key, 0)), self.CG_MAXITER)
        self._cg_floor[key] = trips                                   # survives an invalidation: a later first capture starts from here
        if self._bindings() != self._bound:                           # the eager redo regrew a workspace: every other graph is stale
            if self._graphs:
                self.invalidations += 1
            self._graphs.clear()
            self._bound = self._bindings()
        op = self.den.operator
        ssig = sx.new_full([sx.shape[0]], float(key[0]))
        ssig._kdip_host_value = float(key[0])
        op.set_cg_fixed_trips(trips)
        self.cg_trips[key] = trips
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                so = self.den(sx, ssig)
        finally:
            op.set_cg_fixed_trips(0)
        op.cg_unconverged()            # (the capture pass itself does not run kernels; clear anything the eager redo left)
        self._graphs[key] = (g, sx, so)

    forward = __call__
