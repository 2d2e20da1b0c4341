"""GradientPlan: what one chunk of a minibatch launches.

The reference keys a plan on (shape, dtype, device) once and reuses it
(`operators/cupy/cache.py:32-46`: the cuFFT plan of `_get_nearplane_gradients`'
transforms, `ptycho/solvers/lstsq.py:422-579`).  Here the counterpart of that
plan is the ROUTE a chunk takes through the HIP entries -- which kernels, in
which order, with which workspaces -- decided once per (detector, probe
window, modes, noise model, step rule, mask, switches) and cached on the
operator; `_get_nearplane_gradients` only walks the chunks.  The plan also
names its launches and carries their byte models, so that `bench.py` prices a
launch on what the plan says it moves instead of re-deriving it by entry name.

Routes:
  no_farplane  256^2 (and 512^2 with the fused pass 2): the far plane never
               reaches memory -- forward pass 1 -> column pass + gradient
               factor + inverse pass 1 -> pass 2 + gradients
  split_kept   256^2 / 512^2 with the far plane kept (per-mode poisson steps
               from a stored far plane)
  pos_major    128^2, or 256^2 / 512^2 with a probe window narrower than the
               detector or more modes than the fused pass 2 takes: forward +
               intensity in one position-major kernel, scaled inverse
  pfa          detector sizes p x 2^k, p in {3, 5, 7} (96 ... 3584), gaussian
               model: p x p sub-tiles through the power-of-two register engine
               (csrc/pfa.hip)
  general      every other shape with a mixed-radix plan, gaussian model: the
               three launches of csrc/general.hip
  unfused      what is left (Bluestein sizes, poisson on general shapes): the
               operators one by one on a stored far plane
"""
from dataclasses import dataclass
from types import SimpleNamespace

import torch

from ... import _arrays as A
from ..._lib import check, lib

MODELS = {"gaussian": 0, "poisson": 1}


# ------------------------------------------------------------ byte models
def algorithmic_bytes(name, n, S, det, pw, C, depth=1):
    """HBM bytes the launch `name` must move for n positions (DESIGN.md
    section 3; 0: no model).  T = one position's far plane, P = one probe
    window, D = one pattern."""
    T = 8 * S * det * det
    Tw = 8 * S * pw * det  # the rows of the probe window only (general route)
    P = 8 * pw * pw
    D = 4 * det * det
    box = 8 * (pw + 1) * (pw + 1)
    table = {
        # forward pass 1 hands a far-plane-sized array to the next kernel and
        # stores the object patches
        "tike_fwd_pass1": n * (T + 2 * P + 8) + (S + C) * P,
        # column pass -> intensity -> gradient factor: reads the hand-off and
        # the data, writes the factor
        "tike_fwd_gradient_scale": n * (T + 2 * D),
        # line-search probes: no patches / no gradient factor stored
        "tike_fwd_pass1:cost_only": n * (T + P + 8) + (S + C) * P,
        "tike_fwd_gradient_scale:cost_only": n * (T + D + 4),
        "tike_ptycho_fwd_gradient_scale":
        n * (T + P + 2 * D + 8) + (S + C) * P,
        "tike_ptycho_fwd_intensity": n * (T + P + D + 8) + (S + C) * P,
        "tike_ptycho_fwd_intensity_only": n * (T + P + D + 8) + (S + C) * P,
        "tike_ifft2_crop_scaled": n * (T + S * P + D),
        "tike_grad_ifft2_crop": n * (T + S * P + D),
        # gradient + inverse pass 1: hand-off in, intermediate out
        "tike_grad_ifft2_pass1": n * (2 * T + D),
        # column pass + gradient factor + inverse pass 1 in one launch: the
        # hand-off and the data in, the intermediate out (it reads the
        # hand-off twice; the second read is not algorithmic)
        "tike_fwd_grad_ifft2_pass1": n * (2 * T + D),
        # poisson, every pixel measured: first sweep of the step lengths (the
        # hand-off and the data in), then second sweep + gradient + inverse
        # pass 1 (both in again -- the sweeps are separated by a sum over the
        # whole pattern -- the intermediate out)
        "tike_poisson_steps_grad_ifft2_pass1": n * (3 * T + 2 * D),
        "tike_ifft2_pass2_gradients_scaled": n * (T + 3 * P) + S * P,
        # inverse pass 2 + both gradients: intermediate + patches in,
        # objproj + chi0 out (+ the probe gradient, probe-sized)
        "tike_ifft2_pass2_gradients": n * (T + 3 * P) + S * P,
        # (the launches of all mode groups of a chunk together; the patches are
        # read and objproj rewritten by each of them: + 2 P per further group)
        "tike_ifft2_pass2_gradients_modes": n * (T + 3 * P) + S * P,  # (all groups)
        # the five far-plane-free stages in one call (cgrad's gradient pass:
        # no chi0 stored)
        "tike_lstsq_chunk_gradients":
        (n * (T + 2 * P + 8) + (S + C) * P)
        + (n * (2 * T + D) if det == 256 else n * (T + 2 * D) + n * (2 * T + D))
        + (n * (T + 2 * P) + S * P) + n * (P + box),
        "tike_gradient_scale": n * 3 * D,
        "tike_farplane_gradient": n * (2 * T + D + 4),
        "tike_ifft2_crop": n * (T + S * P),
        "tike_lstsq_gradients": n * (S * P + 2 * P),
        "tike_scatter_patches": n * (P + box),
        "tike_lstsq_step_stats": n * (3 * P + 32),
        # ---- the shape-general route: the hand-offs hold pw rows per tile
        "tike_gen_fwd_rows": n * (Tw + 2 * P + 8) + (S + C) * P,
        "tike_gen_cols_gradient": n * (2 * Tw + D),
        "tike_gen_inv_rows_gradients": n * (Tw + 3 * P) + S * P,
        # ---- the prime-factor route: whole tiles in the sub-tile layout
        "tike_pfa_fwd_gather": n * (T + 2 * P + 8) + (S + C) * P,
        "tike_pfa_fft2": n * 2 * T,
        "tike_pfa_combine_gradient": n * (2 * T + D),
        "tike_pfa_inv_products": n * (T + 3 * P) + S * P,
        # ---- the stages of a multislice object (no patches stored there)
        "tike_fwd_pass1:no_patches": n * (T + P + 8) + (S + C) * P,
        # the probe incident on a slice behind the first: one wave per position in
        "tike_fwd_pass1:incident": n * (2 * T + P + 8),
        "tike_fresnel_colpass": n * 2 * T + D * 2,
        "tike_fft2_pass2_inplace": n * 2 * T,
        # last pass of a slice step + illumination (the wave itself not written)
        "tike_fft2_pass2_intensity": n * (T + D),
        # gradient pass of the last slice: hand-off + data in, one intermediate
        # per slice out
        "tike_fwd_grad_ifft2_pass1_slices": n * (T + D + depth * T),
        # inverse pass 2 + both numerators of a slice: intermediate in, object
        # patch gathered, objproj (+ mode 0 of chi) out
        "tike_ifft2_pass2_products": n * (T + 3 * P) + S * P,
        "tike_ifft2_pass2_products:incident": n * (2 * T + 2 * P),
    }
    return table.get(name, 0)


# ------------------------------------------------------------------ the plan
@dataclass(frozen=True)
class GradientPlan:
    det: int
    pw: int
    S: int
    model: int            # 0 gaussian, 1 poisson
    poisson: bool
    dominant: int         # poisson: step length from the dominant mode only
    all_modes: bool       # poisson: per-mode step lengths
    masked: bool
    route: str
    fused: bool           # pass 2 + gradients in one kernel (chi never stored)
    general: bool         # chi never stored, patches + chi0 stored (general, pfa)
    pfa: bool
    pos_major: bool
    no_farplane: bool
    split_kept: bool
    one_launch: bool      # column pass + gradient factor + inverse pass 1
    steps_in_pass2: bool  # poisson steps applied by pass 2
    chunk: int
    launches: tuple       # the C-ABI entries of one chunk, in order
    groups: tuple = ()    # more modes than one pass-2 launch holds: (first, count)s

    def bytes(self, entry, n, C=0):
        return algorithmic_bytes(entry, n, self.S, self.det, self.pw, C)

    @staticmethod
    def for_(op, S, pw, det, exitwave_options, mask_u8, eigen_modes=0,
             num_eigen=0):
        """The plan of this shape on this operator (cached on it, keyed also on
        the module switches tests and A/B runs flip)."""
        from . import lstsq as L
        eo = exitwave_options
        unmeasured = float(eo.unmeasured_pixels_scaling)
        from ... import _lib
        key = (_lib.DETERMINISTIC, S, pw, det, eo.noise_model,
               eo.step_length_usemodes,
               mask_u8 is not None, unmeasured == 1.0,
               tuple(L.POSITION_MAJOR_SIZES), tuple(L.NO_FARPLANE_SIZES),
               tuple(L.SPLIT_FORWARD_SIZES), tuple(L.ONE_LAUNCH_GRADIENT_SIZES),
               L.POISSON_FROM_HANDOFF, L.POISSON_STEPS_IN_PASS2,
               L.GENERAL_FUSED, L.PFA_ROUTE, L.GENERAL_MIN_DETECTOR,
               L.CHUNK_POSITIONS_OVERRIDE,
               L.mode_groups(S, pw, det, eigen_modes),
               bool(lib.tike_ifft2_pass2_eigen_fits(det, num_eigen,
                                                    eigen_modes)))
        cache = op.__dict__.setdefault("_tike_amd_plans", {})
        if key not in cache:
            cache[key] = GradientPlan._build(S, pw, det, eo, mask_u8, unmeasured,
                                             L, eigen_modes, key[-1])
        return cache[key]

    @staticmethod
    def _build(S, pw, det, eo, mask_u8, unmeasured, L, eigen_modes=0,
               eigen_fits=True):
        poisson = eo.noise_model == "poisson"
        dominant = int(poisson and eo.step_length_usemodes == "dominant_mode")
        all_modes = poisson and not dominant
        pos_major = det in L.POSITION_MAJOR_SIZES
        fused = L.fused_gradients(S, pw, det)
        # 9 ... 16 modes (gaussian): the same far-plane-free kernels -- pass 1
        # and the two-sweep gradient launch take any number of modes -- with
        # the inverse's second pass in two groups of modes
        # (c3m12: 80 -> see profiles/r06_experiments.md section 10)
        groups = () if fused or poisson else L.mode_groups(S, pw, det,
                                                           eigen_modes)
        # (many eigen probes x modes owning them: their slices do not fit the
        # LDS of the fused pass 2 -- chi is stored then)
        fused = (fused or bool(groups)) and eigen_fits
        groups = groups if fused else ()
        # (detector sizes with position-major kernels -- 128, 256, 512 -- keep
        # those for pw < det or many modes: measured faster, c3pad 160 vs 88 k
        # patterns/s, c3m12 69 vs 41 k, profiles/r06_experiments.md)
        general = (not fused
                   and (not pos_major or L.GENERAL_FUSED == "always")
                   and not poisson and L.general_gradients(S, pw, det))
        # ... and among them the sizes p x 2^k, p in {3, 5, 7}, the prime-factor
        # launches (power-of-two register engine on p x p sub-tiles): faster
        # than everything else wherever they apply (96^2 ... 768^2: +15 ... +56 %
        # over the unfused kernels, profiles/r06_experiments.md section 6)
        pfa = general and L.pfa_gradients(S, pw, det)
        # (... except ONE mode below 256 pixels a side, where the unfused
        # kernels are 3-5 % ahead: 96^2 916 vs 949 k patterns/s, 160^2 534 vs
        # 560 k, 192^2 453 vs 469 k; two modes: 96^2 829 vs 794 k, 224^2 246 vs
        # 219 k; one mode from 320^2: +5 ... +36 %)
        if (pfa and S == 1 and det < 256 and L.GENERAL_FUSED != "always"):
            pfa = general = False
        # the LDS line engine pays per work item: below ~256 pixels a side the
        # unfused kernels on the new transforms are faster (45^2: 1360 vs 824 k
        # patterns/s, 64^2: 1117 vs 614 k, 100^2: 559 vs 372 k; 320^2 equal;
        # 384^2 +17 %, 768^2 +16 %, 1024^2 +10 % for the general launches) --
        # and since those three sizes moved to the prime-factor kernels it is
        # nobody's default (lstsq.GENERAL_MIN_DETECTOR)
        if (general and not pfa and det < L.GENERAL_MIN_DETECTOR
                and L.GENERAL_FUSED != "always"):
            general = False
        if general:
            pos_major = False
        # detector sizes with the far-plane-free pipeline (the per-mode poisson
        # steps of 'all_modes' need |F_s|^2: from the forward hand-off with the
        # fused pass 2, from a stored far plane otherwise); 512^2 only together
        # with the fused pass 2
        # (deterministic mode: the hand-off kernels of the per-mode steps sum
        # with float atomics; the stored-far-plane entry does not)
        from ... import _lib
        handoff_steps = L.POISSON_FROM_HANDOFF and not _lib.DETERMINISTIC
        no_farplane = (pos_major
                       and (det in L.NO_FARPLANE_SIZES or (det == 512 and fused))
                       and not (all_modes and not (fused and handoff_steps)))
        split_kept = (pos_major and fused and not no_farplane
                      and det in L.SPLIT_FORWARD_SIZES)
        one_launch = (no_farplane and fused and not poisson
                      and det in L.ONE_LAUNCH_GRADIENT_SIZES)
        # no gradient at unmeasured pixels (none of them, or the default
        # unmeasured_pixels_scaling = 1): the gradient is linear in the step
        # lengths, so their second sweep and the gradient pass are one launch
        # and pass 2 applies them
        steps_in_pass2 = (no_farplane and all_modes and fused and det == 256
                          and (mask_u8 is None or unmeasured == 1.0)
                          and L.POISSON_STEPS_IN_PASS2)
        route = ("pfa" if pfa else "general" if general else
                 "no_farplane" if no_farplane else
                 "split_kept" if split_kept else "pos_major" if pos_major else
                 "unfused")
        launches = {
            "general": ("tike_gen_fwd_rows", "tike_gen_cols_gradient",
                        "tike_gen_inv_rows_gradients"),
            "pfa": ("tike_pfa_fwd_gather", "tike_pfa_fft2",
                    "tike_pfa_combine_gradient", "tike_pfa_fft2",
                    "tike_pfa_inv_products"),
            "no_farplane": ("tike_fwd_pass1",) + (
                ("tike_poisson_steps_grad_ifft2_pass1",) if steps_in_pass2 else
                ("tike_poisson_steps_handoff", "tike_grad_ifft2_pass1")
                if all_modes else ("tike_fwd_grad_ifft2_pass1",) if one_launch
                else ("tike_fwd_gradient_scale",
                      "tike_grad_ifft2_pass1" if fused else
                      "tike_grad_ifft2_crop")) + (
                ("tike_ifft2_pass2_gradients_scaled" if steps_in_pass2 else
                 "tike_ifft2_pass2_gradients",) if fused else
                ("tike_lstsq_gradients",)),
            "split_kept": ("tike_fwd_pass1", "tike_fwd_gradient_scale",
                           "tike_ifft2_pass1_scaled",
                           "tike_ifft2_pass2_gradients"),
            "pos_major": ("tike_ptycho_fwd_intensity", "tike_gradient_scale",
                          "tike_ifft2_pass1_scaled" if fused else
                          "tike_ifft2_crop_scaled",
                          "tike_ifft2_pass2_gradients" if fused else
                          "tike_lstsq_gradients"),
            "unfused": ("tike_ptycho_fwd", "tike_farplane_gradient",
                        "tike_ifft2_crop", "tike_lstsq_gradients"),
        }[route] + ("tike_scatter_patches",)
        return GradientPlan(
            det=det, pw=pw, S=S, model=MODELS[eo.noise_model], poisson=poisson,
            dominant=dominant, all_modes=all_modes, masked=mask_u8 is not None,
            route=route, fused=fused, general=general, pfa=pfa,
            pos_major=pos_major,
            no_farplane=no_farplane, split_kept=split_kept,
            one_launch=one_launch, steps_in_pass2=steps_in_pass2,
            chunk=L.chunk_positions(S, det, pos_major or general),
            launches=tuple("tike_ifft2_pass2_gradients_modes"
                           if groups and e == "tike_ifft2_pass2_gradients" else e
                           for e in launches),
            groups=groups)

    # -------------------------------------------------------- workspaces
    def buffers(self, ws, B, dev, *, varying, want_patches):
        """The chunk workspaces of this route (reused across minibatches)."""
        S, pw, det = self.S, self.pw, self.det
        n = min(self.chunk, max(B, 1))
        c64, f32 = torch.complex64, torch.float32
        b = SimpleNamespace(inten=None, gscale=None, steps=None, unique=None,
                            patches=None, sums=None)
        if want_patches or self.fused or self.general:
            b.patches = ws.get("patches", (max(B, 1), pw, pw), c64, dev)
        b.costs = ws.get("costs", (max(B, 1),), f32, dev)
        if self.pos_major or self.poisson:
            b.inten = ws.get("intensity", (n, det, det), f32, dev)
        if self.pos_major:
            b.gscale = ws.get("gscale", (n, det, det), f32, dev)
        if self.poisson:  # per-(position, mode) step lengths
            b.steps = ws.get("steps", (n, S), f32, dev)
            if self.all_modes and self.no_farplane:
                b.sums = ws.get("poisson_sums", (n, S, 2), f32, dev)
        b.objproj = ws.get("objproj", (n, pw, pw), c64, dev)
        if varying and not self.general:
            b.unique = ws.get("unique", (n, varying, pw, pw), c64, dev)
        # (general: the two hand-offs hold the pw rows of the probe window only;
        # pfa: whole tiles, sub-tile by sub-tile)
        b.far = ws.get("far", (n, 1, S, pw if self.general and not self.pfa
                               else det, det), c64, dev)
        # the inverse transform is out of place (far -> mid); chi is the
        # cropped result and aliases mid when the probe fills the detector
        b.mid = ws.get("mid", tuple(b.far.shape), c64, dev)
        b.chi = b.mid
        if pw != det and not self.general:
            b.chi = ws.get("chi", (n, 1, S, pw, pw), c64, dev)
        # mode 0 of chi is read again after the whole minibatch (step sizes,
        # eigen probes).  When the minibatch is one chunk of a route that
        # stores chi, chi is still intact then and is handed on with a mode
        # stride; otherwise mode 0 is packed.
        b.single_chunk = B <= self.chunk and not self.fused and not self.general
        b.chi0 = (None if b.single_chunk else
                  ws.get("chi0", (max(B, 1), pw, pw), c64, dev))
        return b

    # ---------------------------------------------- forward + far-plane part
    def forward(self, c, k):
        """Everything of chunk k up to the input of the gradient stage: forward
        model, costs, far-plane gradient factor, the inverse's first half (or
        the whole inverse + crop where chi is stored)."""
        getattr(self, "_forward_" + self.route)(c, k)

    def _forward_pfa(self, c, k):
        # gather into p x p sub-tiles -> power-of-two transform of every
        # sub-tile -> p x p combine, cost, gradient factor, inverse combine ->
        # inverse transform; the products in gradients()
        b = c.buf
        from . import lstsq as L
        if L.PFA_SUBTILES_IN_LDS and lib.tike_pfa_fwd_subtiles_supported(
                self.S, self.pw, self.det):
            # 128^2 sub-tiles (384, 640, 896): gathered and transformed inside
            # LDS, written once
            psub = L._workspace(c.op).get(
                "pfa_probe", (self.S + c.C * c.Sm, self.det, self.det),
                torch.complex64, c.psi.device)
            check(
                lib.tike_pfa_fwd_subtiles(
                    A.ptr(c.psi), A.ptr(k.scan), A.ptr(c.probe), A.ptr(c.ep),
                    A.ptr(k.w), c.C, c.Sm, A.ptr(psub), A.ptr(b.mid),
                    A.ptr(k.patches), k.n, self.S, self.pw, self.det, c.H,
                    c.W, c.st), "prime-factor gather + sub-tile transforms")
        else:
            check(
                lib.tike_pfa_fwd_gather(
                    A.ptr(c.psi), A.ptr(k.scan), A.ptr(c.probe), 0, None,
                    A.ptr(c.ep), A.ptr(k.w), c.C, c.Sm, A.ptr(b.far),
                    A.ptr(k.patches), k.n, self.S, self.pw, self.det, c.H,
                    c.W, c.st), "prime-factor gather")
            check(lib.tike_pfa_fft2(A.ptr(b.far), A.ptr(b.mid), k.n * self.S,
                                    self.det, 0, c.st), "sub-tile transforms")
        check(
            lib.tike_pfa_combine_gradient(
                A.ptr(b.mid), A.ptr(k.data_f32()), A.ptr(c.mask_u8),
                A.ptr(k.costs), k.n, self.S, self.det, c.fwd_scale, self.model,
                c.unmeasured, c.nmeasured, 1, c.st),
            "p x p combine + gradient")
        check(lib.tike_pfa_fft2(A.ptr(b.mid), A.ptr(b.far), k.n * self.S,
                                self.det, 1, c.st),
              "inverse sub-tile transforms")

    def _forward_general(self, c, k):
        # K1 rows (patch x probe, zero padding made in LDS) -> K2 columns
        # (intensity, cost, gradient factor, inverse columns) -> K3 in gradients()
        b = c.buf
        check(
            lib.tike_gen_fwd_rows(
                A.ptr(c.psi), A.ptr(k.scan), A.ptr(c.probe), 0, None,
                A.ptr(c.ep), A.ptr(k.w), c.C, c.Sm, A.ptr(b.far),
                A.ptr(k.patches), k.n, self.S, self.pw, self.det, c.H, c.W,
                c.st), "general forward rows")
        check(
            lib.tike_gen_cols_gradient(
                A.ptr(b.far), A.ptr(k.data_f32()), A.ptr(c.mask_u8),
                A.ptr(k.costs), A.ptr(b.mid), k.n, self.S, self.pw, self.det,
                c.fwd_scale, self.model, c.unmeasured, c.nmeasured, c.st),
            "general columns + gradient")

    def _forward_no_farplane(self, c, k):
        # the far-plane waves never reach memory: the forward kernel forms them
        # in registers and leaves the input of its column pass in `far`; the
        # next kernel re-forms them from there, applies the gradient factor
        # and transforms back (factor and costs come out of the same launch)
        b, S, det, pw = c.buf, self.S, self.det, self.pw
        n, st = k.n, c.st
        u16 = int(c.data.dtype == torch.uint16)
        check(
            lib.tike_fwd_pass1(
                A.ptr(c.psi), A.ptr(k.scan), A.ptr(c.probe), 0, None,
                A.ptr(c.ep), A.ptr(k.w), c.C, c.Sm, A.ptr(b.far),
                A.ptr(k.patches) if self.fused else None, n, S, pw, det, c.H,
                c.W, st), "forward pass 1")
        if self.steps_in_pass2:
            check(
                lib.tike_poisson_steps_grad_ifft2_pass1(
                    A.ptr(b.far), A.ptr(k.data), u16, A.ptr(c.mask_u8),
                    A.ptr(k.costs), A.ptr(b.steps), A.ptr(b.sums),
                    A.ptr(b.mid), n, S, det, c.fwd_scale, c.unmeasured,
                    c.nmeasured, c.step_start, c.step_weight, st),
                "poisson step lengths + gradient + inverse pass 1")
            return
        if self.all_modes:
            # gradient factor, costs and the per-mode step lengths from the
            # hand-off: three reads of it, no far plane stored
            check(
                lib.tike_poisson_steps_handoff(
                    A.ptr(b.far), A.ptr(k.data), u16, A.ptr(c.mask_u8),
                    A.ptr(b.gscale), A.ptr(k.costs), A.ptr(b.steps),
                    A.ptr(b.sums), n, S, det, c.fwd_scale, c.unmeasured,
                    c.nmeasured, c.step_start, c.step_weight, st),
                "forward pass 2 + poisson factor and step lengths")
        elif self.one_launch:
            check(
                lib.tike_fwd_grad_ifft2_pass1(
                    A.ptr(b.far), A.ptr(k.data), u16, A.ptr(c.mask_u8),
                    A.ptr(k.costs), A.ptr(b.mid), n, S, det, c.fwd_scale,
                    self.model, c.unmeasured, c.nmeasured, st),
                "column pass + gradient + inverse pass 1")
            return
        else:
            check(
                lib.tike_fwd_gradient_scale(
                    A.ptr(b.far), A.ptr(k.data), u16, A.ptr(c.mask_u8),
                    A.ptr(b.gscale), A.ptr(b.inten) if self.poisson else None,
                    A.ptr(k.costs), None, n, S, det, c.fwd_scale, self.model,
                    c.unmeasured, c.nmeasured, st),
                "forward pass 2 + gradient scale")
        if self.poisson and self.dominant:  # the steps need no far-plane waves
            check(
                lib.tike_poisson_steps(
                    None, A.ptr(b.inten), A.ptr(k.data_f32()),
                    A.ptr(c.mask_u8), A.ptr(b.steps), n, S, det, c.step_start,
                    c.step_weight, 1, st), "poisson step lengths")
        steps = A.ptr(b.steps) if self.poisson else None
        smask = A.ptr(c.mask_u8) if self.poisson else None
        if self.fused:
            check(
                lib.tike_grad_ifft2_pass1(A.ptr(b.far), A.ptr(b.gscale), steps,
                                          smask, S, A.ptr(b.mid), n * S, det,
                                          c.fwd_scale, st),
                "gradient + inverse pass 1")
        else:
            check(
                lib.tike_grad_ifft2_crop(A.ptr(b.far), A.ptr(b.gscale), steps,
                                         smask, S, A.ptr(b.mid), A.ptr(b.chi),
                                         n * S, det, pw, c.fwd_scale,
                                         c.inv_scale, st),
                "gradient + ifft2 + crop")

    def _forward_split_kept(self, c, k):
        # the far plane is kept (per-mode poisson steps from a stored far
        # plane): forward pass 1 -> streamed column pass that stores the
        # far-plane waves (in `mid`) next to the gradient factor -> inverse
        # pass 1 back into `far` -> pass 2 + gradients
        b, S, det, pw = c.buf, self.S, self.det, self.pw
        n, st, d = k.n, c.st, k.data_f32()
        check(
            lib.tike_fwd_pass1(
                A.ptr(c.psi), A.ptr(k.scan), A.ptr(c.probe), 0, A.ptr(k.uq),
                None, A.ptr(k.w), c.C, c.Sm, A.ptr(b.far), A.ptr(k.patches), n,
                S, pw, det, c.H, c.W, st), "forward pass 1")
        check(
            lib.tike_fwd_gradient_scale(
                A.ptr(b.far), A.ptr(d), 0, A.ptr(c.mask_u8), A.ptr(b.gscale),
                A.ptr(b.inten) if self.poisson else None, A.ptr(k.costs),
                A.ptr(b.mid), n, S, det, c.fwd_scale, self.model, c.unmeasured,
                c.nmeasured, st), "forward pass 2 + gradient scale")
        if self.poisson:
            check(
                lib.tike_poisson_steps(
                    A.ptr(b.mid), A.ptr(b.inten), A.ptr(d), A.ptr(c.mask_u8),
                    A.ptr(b.steps), n, S, det, c.step_start, c.step_weight,
                    self.dominant, st), "poisson step lengths")
        check(
            lib.tike_ifft2_pass1_scaled(
                A.ptr(b.mid), A.ptr(b.gscale),
                A.ptr(b.steps) if self.poisson else None,
                A.ptr(c.mask_u8) if self.poisson else None, S, A.ptr(b.far),
                n * S, det, st), "scaled inverse pass 1")

    def _forward_pos_major(self, c, k):
        # forward + intensity in one kernel; the gradient factor is a per-pixel
        # table applied while the inverse transform loads rows
        b, S, det, pw = c.buf, self.S, self.det, self.pw
        n, st, d = k.n, c.st, k.data_f32()
        check(
            lib.tike_ptycho_fwd_intensity(
                A.ptr(c.psi), A.ptr(k.scan), A.ptr(c.probe), 0, A.ptr(k.uq),
                A.ptr(k.w), c.C, c.Sm, A.ptr(b.far), A.ptr(b.inten),
                A.ptr(k.patches) if self.fused else None, n, S, pw, det, c.H,
                c.W, c.fwd_scale, st), "forward + intensity")
        check(
            lib.tike_gradient_scale(A.ptr(b.inten), A.ptr(d), A.ptr(c.mask_u8),
                                    A.ptr(b.gscale), A.ptr(k.costs), n, det,
                                    self.model, c.unmeasured, c.nmeasured, st),
            "gradient scale")
        if self.poisson:
            check(
                lib.tike_poisson_steps(
                    A.ptr(b.far), A.ptr(b.inten), A.ptr(d), A.ptr(c.mask_u8),
                    A.ptr(b.steps), n, S, det, c.step_start, c.step_weight,
                    self.dominant, st), "poisson step lengths")
        if self.fused:
            check(
                lib.tike_ifft2_pass1_scaled(
                    A.ptr(b.far), A.ptr(b.gscale),
                    A.ptr(b.steps) if self.poisson else None,
                    A.ptr(c.mask_u8) if self.poisson else None, S,
                    A.ptr(b.mid), n * S, det, st), "scaled inverse pass 1")
        elif self.poisson:
            check(
                lib.tike_ifft2_crop_scaled_modes(
                    A.ptr(b.far), A.ptr(b.gscale), A.ptr(b.steps),
                    A.ptr(c.mask_u8), S, A.ptr(b.mid), A.ptr(b.chi), n * S,
                    det, pw, c.inv_scale, st), "scaled ifft2 + crop (poisson)")
        else:
            check(
                lib.tike_ifft2_crop_scaled(A.ptr(b.far), A.ptr(b.gscale), S,
                                           A.ptr(b.mid), A.ptr(b.chi), n * S,
                                           det, pw, c.inv_scale, st),
                "scaled ifft2 + crop")

    def _forward_unfused(self, c, k):
        b, S, det, pw = c.buf, self.S, self.det, self.pw
        n, st, d = k.n, c.st, k.data_f32()
        c.op.fwd_device(c.probe, k.scan, c.psi, c.eigen_probe, k.w,
                        out=b.far[:n])
        if self.poisson:
            check(lib.tike_intensity(A.ptr(b.far), A.ptr(b.inten), n, S,
                                     det * det, st), "intensity")
            check(
                lib.tike_poisson_steps(
                    A.ptr(b.far), A.ptr(b.inten), A.ptr(d), A.ptr(c.mask_u8),
                    A.ptr(b.steps), n, S, det, c.step_start, c.step_weight,
                    self.dominant, st), "poisson step lengths")
        check(
            lib.tike_farplane_gradient(
                A.ptr(b.far), A.ptr(d), A.ptr(c.mask_u8), None, A.ptr(k.costs),
                n, S, det, self.model, 1, c.unmeasured, c.nmeasured, st),
            "farplane gradient")
        if self.poisson:
            check(
                lib.tike_scale_modes(A.ptr(b.far), A.ptr(b.steps),
                                     A.ptr(c.mask_u8), n * S, det, st),
                "poisson step scaling")
        check(
            lib.tike_ifft2_crop(A.ptr(b.far), A.ptr(b.mid), A.ptr(b.chi), n * S,
                                det, pw, c.inv_scale, st), "ifft2 + crop")

    # ------------------------------------------------------ gradient stage
    def gradients(self, c, k):
        """Probe gradient, object projection (input of the scatter), mode 0 of
        chi -- from the inverse's second half where chi is never stored, from
        the stored chi otherwise."""
        b, S, det, pw = c.buf, self.S, self.det, self.pw
        n, st = k.n, c.st
        objproj = A.ptr(b.objproj) if c.recover_psi else None
        chi0 = A.ptr(k.chi0) if c.need_chi0 and k.chi0 is not None else None
        if self.pfa:
            check(
                lib.tike_pfa_inv_products(
                    A.ptr(b.far), A.ptr(k.patches), A.ptr(c.probe), 0, None,
                    A.ptr(c.ep), A.ptr(k.w), c.C, c.Sm, objproj, chi0,
                    A.ptr(c.m_probe_update), 1.0 / c.num_batch, n, S, pw, det,
                    c.inv_scale, st), "prime-factor inverse products")
        elif self.general:
            check(
                lib.tike_gen_inv_rows_gradients(
                    A.ptr(b.mid), A.ptr(k.patches), A.ptr(c.probe), 0, None,
                    A.ptr(c.ep), A.ptr(k.w), c.C, c.Sm, objproj, chi0,
                    A.ptr(c.m_probe_update), 1.0 / c.num_batch, n, S, pw, det,
                    c.inv_scale, st), "general inverse rows + gradients")
        elif self.fused:
            # inverse column pass + both gradients + mode 0 of chi, one
            # pixel-major kernel (chi itself never exists in memory)
            p2 = (A.ptr(b.far if self.split_kept else b.mid), A.ptr(k.patches),
                  A.ptr(c.probe), A.ptr(c.ep), A.ptr(k.w), c.C, c.Sm, objproj,
                  chi0, A.ptr(c.m_probe_update), 1.0 / c.num_batch, n, S, det,
                  c.inv_scale)
            if self.groups:
                for first, count in self.groups:
                    check(lib.tike_ifft2_pass2_gradients_modes(
                        *p2, first, count, int(first > 0), st),
                        "inverse pass 2 + gradients (a group of modes)")
            elif self.steps_in_pass2:
                check(lib.tike_ifft2_pass2_gradients_scaled(
                    *p2, A.ptr(b.steps), st),
                    "inverse pass 2 + gradients (x poisson steps)")
            else:
                check(lib.tike_ifft2_pass2_gradients(*p2, st),
                      "inverse pass 2 + gradients")
        else:
            # one pass over chi: probe gradient, object projection, patches
            check(
                lib.tike_lstsq_gradients(
                    A.ptr(b.chi), A.ptr(k.scan), A.ptr(c.psi), A.ptr(c.probe),
                    A.ptr(c.ep), A.ptr(k.w), c.C, c.Sm,
                    None,  # on the fly: L2-resident
                    A.ptr(k.patches), A.ptr(c.m_probe_update), objproj, n, S,
                    pw, c.H, c.W, st), "probe gradient + object projection")
