"""Denoising driver around the HIP attention path (reference loop: paint_with_words/paint_with_words.py
:431-506; inpaint variant paint_with_words_inpaint.py:230-266).

Three execution modes over the SAME arithmetic:
  "eager"   two batch-1 UNet calls per step (cond dict, then uncond dict), exactly the reference's
            call pattern (:483-499) -- the parity mode.
  "folded"  cond and uncond rows of all images in ONE UNet call ([cond rows..., uncond rows...]); the
            PwW bias is gated per row by the `_PWW_ROW_GATE` coefficients (kernel arg bias_coeff), and
            the qk reductions stay per image (QKProxy).
  "graph"   "folded", with the UNet call captured ONCE into a hipGraph (torch.cuda.CUDAGraph on ROCm) and
            replayed for every step of every request of the same geometry: at batch 1 the UNet is
            launch-bound (~1.5k kernels per call), so replay removes the host launch cost. The only
            step-dependent kernel arguments -- the Python scalars c0 * g(sigma) the user's
            weight_function produces -- live in device words (attention.CoeffSlots) that the host
            rewrites before each replay. A weight function that is not of the form
            c * w * reduce(qk) bakes sigma into torch ops; the sampler then keeps one graph per step.
"""
import torch

from . import ops
from .attention import install, refresh_kv_cache, refresh_orig_cache, ROW_GATE, KV_CACHE, COEFF_SLOTS, CoeffSlots, COMPACT_W, COMPACT_IDX, GATED_ROWS
from .conditioning import PwWContext

ORIG = "CROSS_ATTENTION_WEIGHT_ORIG"


def _orig_pending(d):
    return isinstance(d, PwWContext) and d.pending(ORIG)


def initial_latents(seed, in_channels, height, width, region_masks=None, extra_seeds=None, batch_seeds=None):
    """:445-455. CPU generator, exactly as the reference, so results do not depend on the device or
    the number of GPUs. `region_masks`: callable(dtype, size) -> list of [1,1,h,w] masks for the seeded
    regions (the reference's _get_binary_mask). `batch_seeds`: one latent per seed (batched mode)."""
    seeds = [seed] if batch_seeds is None else list(batch_seeds)
    size = (1, in_channels, height // 8, width // 8)
    out = []
    for s in seeds:
        latents = torch.randn(size, generator=torch.manual_seed(s))
        if extra_seeds:
            print("Use region based seeding: ", extra_seeds)
            multi = [torch.randn(size, generator=torch.manual_seed(_s)) for _s in extra_seeds.values()]
            masks = region_masks(latents[0].dtype, size[-2:])
            foreground = (sum(masks) > 0).squeeze()
            summed = sum(l * m for l, m in zip(multi, masks))
            latents[:, :, foreground] = summed[:, :, foreground]
        out.append(latents)
    return torch.cat(out, dim=0)


def _as_list(x, n):
    """One context dict per image: a single dict serves every image of the batch."""
    if isinstance(x, (list, tuple)):
        if len(x) not in (1, n):
            raise ValueError("got %d context dicts for %d images" % (len(x), n))
        return list(x) if len(x) == n else [x[0]] * n
    return [x] * n


def _fold_context(cond, uncond, n_images, device):
    """Build the single dict of a folded call: rows [cond x n, uncond x n]. `cond` / `uncond`: one dict, or one dict per
    image (paint_with_words_batch: per-image prompts and color maps). Per-image weight maps are stacked to
    [2n, 1, N, 77] -- the kernels take the image axis through bias_stride[0] -- with zeros for the unconditional rows
    (their gate is 0, the values are never used); a shared map stays [N, 77]."""
    conds, unconds = _as_list(cond, n_images), _as_list(uncond, n_images)
    shared = all(c is conds[0] for c in conds)

    def rows(dicts):
        if all(d is dicts[0] for d in dicts):
            t = dicts[0]["CONTEXT_TENSOR"]
            return t.expand(n_images, -1, -1) if t.shape[0] == 1 else t
        return torch.cat([d["CONTEXT_TENSOR"] for d in dicts], dim=0)

    folded = conds[0].copy() if isinstance(conds[0], PwWContext) else PwWContext(conds[0])      # (keeps a pending ORIG map pending)
    folded[KV_CACHE] = {}
    folded.pop("_PWW_ORIG_CACHE", None)
    folded["CONTEXT_TENSOR"] = torch.cat([rows(conds), rows(unconds)], dim=0).contiguous()
    folded[ROW_GATE] = torch.cat([torch.ones(n_images), torch.zeros(n_images)]).to(device=device, dtype=torch.float32)
    folded[GATED_ROWS] = n_images        # what the gate holds, for the host side (work distribution of the fused launch)
    if not shared:
        if any(_orig_pending(c) for c in conds):
            # per-image fallback maps, stacked on first use: [2n, H, W, 77] (zeros for the unconditional rows)
            def stacked_orig(conds=conds):
                w = torch.stack([c[ORIG] for c in conds], dim=0)
                return torch.cat([w, torch.zeros_like(w)], dim=0)
            folded.pop(ORIG, None)
            folded.set_lazy(ORIG, stacked_orig)
        for key in [k for k in dict.keys(conds[0]) if k.startswith("CROSS_ATTENTION_WEIGHT_") and not (k == ORIG and ORIG in folded._thunks)]:
            maps = [c[key] for c in conds]
            if not all(torch.is_tensor(m) for m in maps):
                raise ValueError("per-image contexts must all carry a tensor for %s" % key)
            w = torch.stack(maps, dim=0)                       # [n, N, 77]  (ORIG: [n, H, W, 77])
            w = torch.cat([w, torch.zeros_like(w)], dim=0)
            folded[key] = w if key.endswith("_ORIG") else w.unsqueeze(1)
        # private hints: the column bound is the largest of the images'; the compact forms are padded to one width
        # (unused slots: column index -1) and stacked like the maps
        folded["_PWW_BIAS_COLS"] = max((int(c.get("_PWW_BIAS_COLS", 0) or 0) for c in conds), default=0) if all(c.get("_PWW_BIAS_COLS") for c in conds) else 0
        idxs = [c.get(COMPACT_IDX) for c in conds]
        for key in [k for k in list(folded) if k.startswith(COMPACT_W) or k == COMPACT_IDX]:
            folded.pop(key)
        if all(torch.is_tensor(i) for i in idxs):
            R = max(int(i.numel()) for i in idxs)
            pad_i = [torch.cat([i, i.new_full((R - i.numel(),), -1)]) for i in idxs]
            idx = torch.stack(pad_i + [torch.full_like(pad_i[0], -1)] * n_images, dim=0)          # [2n, R]
            ok = True
            stacked = {}
            for key in [k for k in conds[0] if k.startswith(COMPACT_W)]:
                ws = [c.get(key) for c in conds]
                if not all(torch.is_tensor(w) for w in ws):
                    ok = False
                    break
                ws = [torch.cat([w, w.new_zeros(w.shape[0], R - w.shape[1])], dim=1) for w in ws]
                w = torch.stack(ws, dim=0)                                                         # [n, N, R]
                stacked[key] = torch.cat([w, torch.zeros_like(w)], dim=0).contiguous()
            if ok:
                folded.update(stacked)
                folded[COMPACT_IDX] = idx.contiguous()
    return folded


def weight_function_signature(f):
    """Cache key of the FALL-BACK path only (weight functions that are not c * w * reduce(qk): one hipGraph per step with
    sigma baked into torch ops). What such graphs depend on in the function: its code and every constant it can see --
    closure cells, defaults, the constants of nested code objects (inner lambdas, comprehensions), numeric globals. A function
    that reaches anything else (a non-numeric global such as a helper function or a config object, a closure over a mutable
    object, a callable object) is keyed by a fresh token, i.e. it never matches a cached key and is re-captured: slower, never
    stale. (The regular path needs none of this: the host re-evaluates the function every step, attention.CoeffSlots.)"""
    fresh = ("fresh", object())

    def atom(v):
        if isinstance(v, (int, float, str, bool, bytes, type(None))):
            return ("v", type(v).__name__, v)
        if isinstance(v, tuple) and all(isinstance(x, (int, float, str, bool, type(None))) for x in v):
            return ("t", v)
        return None

    def code_key(code):
        consts = []
        for c in code.co_consts:
            if hasattr(c, "co_code"):
                consts.append(code_key(c))
            else:
                a = atom(c)
                consts.append(a if a is not None else ("repr", repr(c)))     # frozenset / Ellipsis ...: immutable constants
        return ("code", code.co_code, tuple(consts), code.co_names, code.co_varnames[:code.co_argcount])

    def names_of(code):
        out = set(code.co_names)
        for c in code.co_consts:
            if hasattr(c, "co_code"):
                out |= names_of(c)
        return out

    code = getattr(f, "__code__", None)
    if code is None:       # callable object / builtin
        return fresh
    cells = tuple(atom(c.cell_contents) for c in (f.__closure__ or ()))
    dvals = tuple(atom(v) for v in (f.__defaults__ or ()))
    kvals = tuple((k, atom(v)) for k, v in sorted((f.__kwdefaults__ or {}).items()))
    if any(c is None for c in cells) or any(d is None for d in dvals) or any(v is None for _, v in kvals):
        return fresh
    defaults = dvals + kvals
    import types
    g = getattr(f, "__globals__", {})
    globs = []
    for n in sorted(names_of(code)):       # (co_names also holds attribute names -- log, max, std ...: only real globals count)
        if n in g:
            a = atom(g[n])
            if a is None and not isinstance(g[n], types.ModuleType):
                return fresh       # helper function, config object ...: cannot be proven constant
            globs.append((n, a if a is not None else ("module", g[n].__name__)))
    return (code_key(code), cells, defaults, tuple(globs))


class _GraphedUNet:
    """Capture-once / replay-many wrapper of the folded UNet call: key "all" (one graph for every step) while the weight
    function's scalars travel in device words, else one graph per step index."""

    def __init__(self, unet):
        self.unet = unet
        self.graphs = {}
        self.pool = None
        self.static_in = None
        self.static_t = None
        self.captures = 0

    def reset(self):
        """Drop every captured graph AND the memory pool they shared: once the last graph of a pool is gone the
        allocator retires the pool, and capturing into the stale handle trips an internal assert."""
        self.graphs.clear()
        self.pool = None

    def __call__(self, step, x, t_value, context):
        if self.static_in is None or self.static_in.shape != x.shape or self.static_in.dtype != x.dtype:
            self.reset()
            self.static_in = torch.empty_like(x)
            self.static_t = torch.zeros((), dtype=torch.float32, device=x.device)
        self.static_in.copy_(x)
        self.static_t.fill_(float(t_value))
        slots = context.get(COEFF_SLOTS)
        key = "all" if slots is not None and not slots.unsupported else step
        entry = self.graphs.get(key)
        if entry is None:
            # warm-up on a side stream (required before capture; also the discovery pass of the coefficient slots), then capture
            if slots is not None:
                slots.discover = True
                slots.begin_forward()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self.unet(self.static_in, self.static_t, encoder_hidden_states=context)
            torch.cuda.current_stream().wait_stream(s)
            if slots is not None:
                slots.discover = False
                slots.begin_forward()
                key = "all" if not slots.unsupported else step
            g = torch.cuda.CUDAGraph()
            if self.pool is None:
                self.pool = torch.cuda.graph_pool_handle()
            with torch.cuda.graph(g, pool=self.pool):
                out = self.unet(self.static_in, self.static_t, encoder_hidden_states=context).sample
            entry = (g, out)
            self.graphs[key] = entry
            self.captures += 1
        g, out = entry
        g.replay()
        return out


class PwWSampler:
    """Runs the reference's sampling loop for one or many images on one GPU."""

    def __init__(self, unet, scheduler, mode="eager"):
        if mode not in ("eager", "folded", "graph"):
            raise ValueError("mode must be eager | folded | graph")
        self.unet, self.scheduler, self.mode = unet, scheduler, mode
        install(unet)
        self._graphed = _GraphedUNet(unet) if mode == "graph" else None
        self._graph_sig = None
        self._fallback_sig = None
        self._static_folded = None
        self._request_folded = None
        self._scratch_modules = None
        self._errors = ops.FusedErrorWatch()
        self.handoff_pending = False     # the last request issued launches with an in-kernel hand-off (PWW_QPROJ_STAT=0 / shapes the
                                         # to_q GEMM does not cover): its latents must not reach a caller before check_errors() has looked

    def _static_context(self, folded, weight_function, latents, timesteps):
        """Captured graphs read the context tensors by ADDRESS: keep one set of static tensors alive and copy each new
        request's values into them; rebuild the graphs when the geometry changes. The weight function and the schedule are
        NOT part of the key while the function's scalars travel in device words (CoeffSlots) -- fresh lambdas, other
        constants, another step count all replay the same graph; only the per-step fall-back keys them."""
        def tensor_sig(d):      # (the full-resolution fallback map is not part of the geometry: it exists only once a layer asked for it)
            return tuple(sorted((k, tuple(v.shape), v.dtype) for k, v in d.items() if torch.is_tensor(v) and k != ORIG))
        sig = (tuple(latents.shape), tensor_sig(folded), tuple(sorted((k, v) for k, v in folded.items() if isinstance(v, int) and not isinstance(v, bool))))
        self._request_folded = folded
        if sig != self._graph_sig:
            self._graphed.reset()
            self._graph_sig = sig
            self._fallback_sig = None
            self._static_folded = PwWContext({k: (v.clone() if torch.is_tensor(v) else v) for k, v in folded.items() if k != ORIG})
            # CROSS_ATTENTION_WEIGHT_ORIG stays pending in the static context too: the first layer that takes the fallback (in the
            # eager warm-up pass before the capture) clones the CURRENT request's map; later requests copy theirs into it
            self._static_folded.set_lazy(ORIG, lambda: self._request_folded[ORIG].clone())
            self._static_folded[COEFF_SLOTS] = CoeffSlots(latents.device)
        else:
            for k, v in folded.items():
                if torch.is_tensor(v) and k != ORIG:
                    self._static_folded[k].copy_(v)
            if dict.__contains__(self._static_folded, ORIG) and torch.is_tensor(self._static_folded[ORIG]):
                self._static_folded[ORIG].copy_(folded[ORIG])      # (builds this request's map: the captured layers read it)
            refresh_kv_cache(self._static_folded)     # new prompt embedding -> new K|V, same addresses
            refresh_orig_cache(self._static_folded)   # new color map -> new fallback maps, same addresses
        slots = self._static_folded[COEFF_SLOTS]
        if slots.unsupported:      # one graph per step: those do depend on the function's constants and on the schedule
            fsig = (weight_function_signature(weight_function), tuple(float(t) for t in timesteps))
            if fsig != self._fallback_sig:
                self._graphed.reset()
                self._fallback_sig = fsig
        return self._static_folded

    def _sigma_and_index(self, i, t):
        sch = self.scheduler
        ts = sch.timesteps
        # the reference looks the index up by timestep equality (:473); PLMS repeats a timestep, so a
        # scheduler that is not 1:1 falls back to the loop counter (SURVEY.md 8 a-note)
        hits = (ts == t).nonzero()
        idx = int(hits.item()) if hits.numel() == 1 else i
        return sch.sigmas[idx], idx

    @torch.no_grad()
    def sample(self, cond, uncond, latents, timesteps, guidance_scale, weight_function, extra_channels=None,
               on_step=None):
        """cond / uncond: the two context dicts of the PwW protocol, or one dict per image (per-image prompts / maps).
        latents: [n_images, C, h, w] already scaled by init_noise_sigma (or noised for img2img).
        extra_channels: inpaint's cat([mask, masked_image_latents]) ([n or 1, 5, h, w]) or None."""
        sch, unet, dev = self.scheduler, self.unet, latents.device
        self._errors.poll(wait=False)      # a hand-off time-out of an earlier request surfaces here (or in check_errors())
        install(unet)      # the plug is a class-level patch (:193-195): put it back if somebody removed it since __init__
        n = latents.shape[0]
        udt = unet.dtype if hasattr(unet, "dtype") else next(unet.parameters()).dtype
        conds, unconds = _as_list(cond, n), _as_list(uncond, n)
        if self.mode == "eager":   # per-request caches of the fused K|V projections (prompt constant over the steps)
            for d in conds + unconds:
                d.setdefault(KV_CACHE, {}).clear()
        folded = None
        if self.mode != "eager":
            folded = _fold_context(cond, uncond, n, dev)
            if self._graphed is not None:
                folded = self._static_context(folded, weight_function, latents, timesteps)
        if extra_channels is not None and extra_channels.shape[0] != n:
            extra_channels = extra_channels.expand(n, -1, -1, -1)
        for i, t in enumerate(timesteps):
            sigma, _ = self._sigma_and_index(i, t)
            x = sch.scale_model_input(latents, t)
            if extra_channels is not None:
                x = torch.cat([x, extra_channels.to(x.dtype)], dim=1)
            if self.mode == "eager":
                eps_c, eps_u = [], []
                for j in range(n):   # the reference is batch-1 (:445); images are independent
                    conds[j].update({"SIGMA": sigma, "WEIGHT_FUNCTION": weight_function})
                    eps_c.append(unet(x[j:j + 1], t, encoder_hidden_states=conds[j]).sample)
                    unconds[j].update({"SIGMA": sigma, "WEIGHT_FUNCTION": lambda w, sigma, qk: 0.0})
                    eps_u.append(unet(x[j:j + 1], t, encoder_hidden_states=unconds[j]).sample)
                eps_c, eps_u = torch.cat(eps_c), torch.cat(eps_u)
            else:
                folded.update({"SIGMA": sigma, "WEIGHT_FUNCTION": weight_function})
                x2 = torch.cat([x, x], dim=0).to(udt)
                if self.mode == "graph":
                    slots = folded[COEFF_SLOTS]
                    if not slots.unsupported and not slots.update(weight_function, sigma):
                        # the function changed its structure (another statistic, not symbolic any more): capture again
                        self._graphed.reset()
                        slots.reset()
                    out = self._graphed(i, x2, float(t), folded)
                else:
                    out = unet(x2, t, encoder_hidden_states=folded).sample
                eps_c, eps_u = out[:n], out[n:]
            if eps_c.dtype in (torch.float16, torch.bfloat16):
                noise_pred = ops.cfg_combine(eps_c, eps_u, guidance_scale)     # fp32, :501-503
            else:
                noise_pred = eps_u + guidance_scale * (eps_c - eps_u)
            latents = sch.step(noise_pred, t, latents).prev_sample
            if on_step is not None:
                on_step(i, t, latents)
        # did any fused cross-attention launch of this request time out in its hand-off? One 4-byte device -> host copy behind an
        # event (ops.FusedErrorWatch): looked at when the next request starts and by check_errors() -- never a host stall here
        if self._scratch_modules is None:
            self._scratch_modules = [m for m in unet.modules() if m.__class__.__name__ in ("CrossAttention", "Attention")]
        self.handoff_pending = self._errors.post(self._scratch_modules)
        return latents

    def check_errors(self):
        """Wait for the requests issued so far and raise PwwHipError if a fused cross-attention hand-off of any of them timed out
        (their outputs are NaN). The PIL-returning entry points call this after decoding; the latent-returning ones
        (`return_latents=True`) call `checked()`."""
        self._errors.poll(wait=True)
        self.handoff_pending = False

    def checked(self, latents):
        """`latents` of the request just issued, safe to hand to a caller: if that request contained launches with an in-kernel
        hand-off (the only launches that can fail at run time) this waits for it and raises on a time-out -- a single call never
        hands back NaN latents silently. The default path (statistic formed in the to_q GEMM, no hand-off) has nothing to wait
        for and stays asynchronous."""
        if self.handoff_pending:
            self.check_errors()
        return latents
