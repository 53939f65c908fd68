"""denoise-steps/sec of the CTSD SD-3.5 MMDiT hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps 5 --warmup 2          (N > 1: re-launches itself, one rank per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot loop of ctsd.py:1496-1575 over one synthetic sample:
model forward at the CFG batch (2 x [16 frames x 6 views x 16 x 32 x 56] latents, 154 text
tokens) + guidance combine + FlowMatch-Euler update, weights random-initialised (no
checkpoints offline), everything resident in HBM when the clock starts.  With N > 1 every rank
denoises its own sample (replicas; the path has no exchange step), so
value = N*K / max-over-ranks time ("weak" scaling).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0       # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBPS = 8000.0          # HBM3E (same guide; 6.29 TB/s measured copy)

# model kwargs of examples/ctsd_35_6views_video_generation.json:45-107 (reference repo)
MODEL_KWARGS = dict(
    dual_attention_layers=list(range(13)), attention_head_dim=64, caption_projection_dim=1536,
    in_channels=16, joint_attention_dim=4096, num_attention_heads=24, num_layers=24, out_channels=16,
    patch_size=2, pooled_projection_dim=2048, pos_embed_max_size=384, qk_norm="rms_norm",
    qk_norm_on_additional_modules="rms_norm", sample_size=128, perspective_modeling_type="implicit",
    projection_class_embeddings_input_dim=2816, enable_crossview=True, crossview_attention_type="rowwise",
    crossview_block_layers=[1, 5, 9, 13, 17, 21], crossview_gradient_checkpointing=True,
    enable_temporal=True, temporal_attention_type="rowwise",
    temporal_block_layers=[2, 3, 6, 7, 10, 11, 14, 15, 18, 19, 22, 23],
    temporal_gradient_checkpointing=True, mixer_type="AlphaBlender", merge_factor=2)
WORKLOAD = dict(B=1, T=16, V=6, C=16, H=32, W=56, text_len=154, guidance_scale=4.0, inference_steps=40)


def synth_init_(model, seed: int):
    """Seeded synthetic weights on the device (the rule of oracle.synth_param)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("mix_factor"):
                p.fill_(2.0)
            elif p.dim() == 1:
                p.normal_(0.0, 0.05, generator=g)
                if name.endswith(".weight"):
                    p.add_(1.0)
            else:
                fan_in = p[0].numel()
                std = fan_in ** -0.5
                if ".norm1.linear" in name or ".norm1_context.linear" in name or name.startswith("norm_out.linear"):
                    std *= 0.5
                p.normal_(0.0, std, generator=g)


def build_model(kwargs, dev, seed=0):
    from opendwm_amd.dit import DiTCrossviewTemporalConditionModel
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device(dev):
            model = DiTCrossviewTemporalConditionModel(**kwargs)
    finally:
        torch.set_default_dtype(old)
    model = model.to(device=dev, dtype=torch.bfloat16)
    synth_init_(model, seed)
    return model.eval()


def variant_kwargs(layout: bool) -> dict:
    """constructor kwargs of the two models of BASELINE config 3: text+layout (ImageAdapter + point-wise temporal
    attention, 13 added time ids; examples/ctsd_35_df16_6views_video_generation_with_layout.json) or text only"""
    kwargs = dict(MODEL_KWARGS)
    if layout:
        kwargs.update(temporal_attention_type="pointwise", projection_class_embeddings_input_dim=3328,
                      condition_image_adapter_config=dict(in_channels=6, channels=[1536] * 6,
                                                          is_downblocks=[True] + [False] * 5, num_res_blocks=2,
                                                          downscale_factor=8, use_zero_convs=True))
    return kwargs


def make_conditions(dev, seed, w=WORKLOAD, n_time_ids=None, layout=False):
    n_time_ids = n_time_ids or (13 if layout else 11)
    cond = _make_conditions(dev, seed, w, n_time_ids)
    if layout:          # 3dbox + hdmap condition images (6 channels) at pixel resolution, CFG-doubled like the rest
        gl = torch.Generator(device="cuda").manual_seed(77 + seed)
        cond["condition_image_tensor"] = torch.rand(2 * w["B"], w["T"], w["V"], 6, 8 * w["H"], 8 * w["W"], device=dev,
                                                    generator=gl).to(torch.bfloat16)
    return cond


def _make_conditions(dev, seed, w, n_time_ids, text_dim=4096, pooled_dim=2048):
    g = torch.Generator(device="cuda").manual_seed(1000 + seed)
    B2, T, V = 2 * w["B"], w["T"], w["V"]
    ring = torch.zeros(V, V, dtype=torch.bool)
    for i in range(V):
        for d in (-1, 0, 1):
            ring[i, (i + d) % V] = True
    return dict(
        encoder_hidden_states=(torch.randn(B2, T, V, w["text_len"], text_dim, device=dev, generator=g) * 0.1).to(torch.bfloat16),
        pooled_projections=(torch.randn(B2, T, V, pooled_dim, device=dev, generator=g) * 0.1).to(torch.bfloat16),
        disable_crossview=torch.zeros(B2, dtype=torch.bool, device=dev),
        disable_temporal=torch.zeros(B2, dtype=torch.bool, device=dev),
        crossview_attention_mask=ring[None].repeat(B2, 1, 1).to(dev),
        added_time_ids=torch.rand(B2, T, V, n_time_ids, device=dev, generator=g) * 2 - 1,
    )


class ClockPowerSampler:
    """GPU clock and socket power DURING the timed region (the step runs at the board's power limit, so the sustained clock - not the
    2.4 GHz the nominal MFMA peak assumes - is what bounds it): a thread reads the amdgpu sysfs files of the device (current sclk level
    of pp_dpm_sclk, hwmon power1_average / power1_input) every `period` seconds; where they are not readable it falls back to one
    `rocm-smi --showclocks --showpower --json` per second.  Host-side only: nothing is launched on the GPU."""

    def __init__(self, device_index: int = 0, period: float = 0.1):
        import threading
        self.period, self.samples, self._stop, self.source, self.error = period, [], threading.Event(), None, None
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._dir = self._find_sysfs(device_index)
        self.index = device_index

    @staticmethod
    def _find_sysfs(index: int):
        import glob
        try:
            p = torch.cuda.get_device_properties(index)
            want = f"{int(p.pci_domain_id):04x}:{int(p.pci_bus_id):02x}:{int(p.pci_device_id):02x}"
        except Exception:
            want = None
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
        for c in cards:
            try:
                if want is not None and os.path.basename(os.path.realpath(c)).startswith(want) and os.path.exists(c + "/pp_dpm_sclk"):
                    return c
            except Exception:
                pass
        amd = [c for c in cards if os.path.exists(c + "/pp_dpm_sclk")]
        return amd[index] if index < len(amd) else (amd[0] if amd else None)

    def _read_sysfs(self):
        import glob
        mhz = watts = None
        for ln in open(self._dir + "/pp_dpm_sclk"):
            if "*" in ln:
                mhz = float(ln.split(":")[1].strip().split("Mhz")[0].split("MHz")[0])
        for name in ("power1_average", "power1_input"):
            for f in glob.glob(self._dir + "/hwmon/hwmon*/" + name):
                try:
                    watts = float(open(f).read()) * 1e-6
                    break
                except Exception:
                    pass
            if watts is not None:
                break
        return mhz, watts

    def _read_smi(self):
        import subprocess
        out = subprocess.run(["rocm-smi", "-d", str(self.index), "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = next(iter(json.loads(out).values()))
        mhz = watts = None
        for k, v in d.items():
            kl = k.lower()
            if "sclk" in kl and "clock speed" in kl:
                mhz = float(str(v).strip("()").lower().replace("mhz", ""))
            if "power" in kl and "(w)" in kl and watts is None:
                watts = float(v)
        return mhz, watts

    def _run(self):
        use_smi = self._dir is None
        while not self._stop.is_set():
            try:
                if not use_smi:
                    mhz, watts = self._read_sysfs()
                    self.source = "sysfs " + self._dir
                    if mhz is None and watts is None:
                        use_smi = True
                        continue
                else:
                    mhz, watts = self._read_smi()
                    self.source = "rocm-smi"
                self.samples.append((time.perf_counter(), mhz, watts))
            except Exception as e:
                self.error = repr(e)
                if use_smi:
                    return
                use_smi = True
            self._stop.wait(1.0 if use_smi else self.period)

    def start(self):
        self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        self._thread.join(timeout=15)

    def summary(self, t0: float, t1: float):
        """mean / min / max over the samples taken inside [t0, t1] (perf_counter seconds)"""
        rs = [r for r in self.samples if t0 <= r[0] <= t1]
        out = {"source": self.source, "samples": len(rs), "period_s": self.period, "error": self.error}
        for name, col in (("sclk_mhz", 1), ("socket_power_w", 2)):
            v = [r[col] for r in rs if r[col] is not None]
            out[name] = None if not v else {"mean": sum(v) / len(v), "min": min(v), "max": max(v)}
        if out["sclk_mhz"]:
            out["peak_at_sustained_clock_tflops"] = PEAK_BF16_TFLOPS * out["sclk_mhz"]["mean"] / 2400.0
        return out


class KernelTimer:
    """HIP-event bracket around every dwm_gemm_bf16 / dwm_attention_fwd launch, recorded on the
    stream the kernel is launched on; durations are read after the timed region."""

    def __init__(self):
        self.records = []           # (kind, flops, start_event, end_event)
        self.small_bytes = 0.0      # algorithmic bytes of the short-sequence (packed kernel) attention launches
        self.group_bytes = 0.0      # the same for the group-masked (cross-view) launches
        self.shapes = []            # per GEMM record: (M, N, K, epilogue, implicit conv)
        self.four_wave = []         # per GEMM record: the launch was served by gemm4w_kernel (gemm_bf16_4w.hip), not gemm_bf16_kernel
        self.enabled = False

    def install(self):
        from opendwm_amd import _lib, ops
        gemm0, attn0 = ops.gemm, ops.attention
        self._orig = (gemm0, attn0)
        timer = self
        count4w = _lib.load().dwm_gemm4w_launches        # host-side launch counter of the 4-wave kernels

        def gemm(a, w, *args, **kw):
            if not timer.enabled:
                return gemm0(a, w, *args, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st = torch.cuda.current_stream()
            s.record(st)
            n4 = count4w()
            out = gemm0(a, w, *args, **kw)
            e.record(st)
            timer.four_wave.append(count4w() > n4)
            grid = kw.get("a_grid")
            M = (grid.pixels // 4 if kw.get("stride2") else grid.pixels) if grid is not None else (kw.get("rows") or a.shape[0])
            timer.records.append(("gemm", 2.0 * M * w.shape[0] * w.shape[1], s, e))
            timer.shapes.append((M, w.shape[0], w.shape[1], kw.get("epilogue", 0), a.shape[1] != w.shape[1]))
            return out

        def attention(q, k, v, out, rowmap, heads, **kw):
            if not timer.enabled:
                return attn0(q, k, v, out, rowmap, heads, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st = torch.cuda.current_stream()
            s.record(st)
            attn0(q, k, v, out, rowmap, heads, **kw)
            e.record(st)
            L = rowmap.L0 + (kw["q1"].shape[0] // rowmap.n_problems if kw.get("q1") is not None else 0)
            fl = 4.0 * rowmap.n_problems * heads * L * L * 64
            small = L <= 32 and kw.get("q1") is None and kw.get("group_mask") is None and kw.get("dense_mask") is None
            # attn_group_lds_kernel: group-masked problems of G whole groups of 8..32 tokens (row-wise cross-view attention)
            grouped = (kw.get("group_mask") is not None and kw.get("q1") is None and kw.get("lse") is None and
                       8 <= rowmap.group_size <= 32 and kw["group_mask"].shape[-1] * rowmap.group_size == L)
            timer.records.append(("attn_small" if small else "attn_group" if grouped else "attn", fl, s, e))
            if small:           # attn_small_kernel is HBM-bound: q, k, v read + o written once
                timer.small_bytes += 4.0 * rowmap.n_problems * L * heads * 64 * 2
            if grouped:
                timer.group_bytes += 4.0 * rowmap.n_problems * L * heads * 64 * 2

        ops.gemm, ops.attention = gemm, attention
        import opendwm_amd.blocks as blocks
        import opendwm_amd.dit as dit
        for mod in (blocks, dit):
            mod.ops = ops
        return self

    def uninstall(self):
        from opendwm_amd import ops
        ops.gemm, ops.attention = self._orig

    def shape_table(self, top=25):
        """per-shape totals of the recorded GEMM launches, slowest first (diagnostics: `--gemm-shapes`)"""
        agg = {}
        gemms = [(f, s.elapsed_time(e)) for k, f, s, e in self.records if k == "gemm"]
        for key, (f, ms) in zip(self.shapes, gemms):
            a = agg.setdefault(key, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += f
            a[2] += ms
        rows = sorted(agg.items(), key=lambda kv: -kv[1][2])[:top]
        return [dict(M=k[0], N=k[1], K=k[2], epilogue=k[3], conv=bool(k[4]), launches=v[0], ms=round(v[2], 3),
                     tflops=round(v[1] / v[2] / 1e9, 1)) for k, v in rows]

    def summary(self):
        out = {}
        for kind in ("gemm", "attn", "attn_small", "attn_group"):
            rs = [(f, s.elapsed_time(e)) for k, f, s, e in self.records if k == kind]
            if rs:
                fl, ms = sum(f for f, _ in rs), sum(t for _, t in rs)
                out[kind] = dict(launches=len(rs), flops=fl, ms=ms, tflops=fl / ms / 1e9,
                                 avg_us=1e3 * ms / len(rs))
        # the GEMM launches by the kernel that served them
        if "gemm" in out and len(self.four_wave) == out["gemm"]["launches"]:
            gemms = [(f, s.elapsed_time(e)) for k, f, s, e in self.records if k == "gemm"]
            split = {}
            for name, want in (("gemm4w_kernel", True), ("gemm_bf16_kernel", False)):
                rs = [r for r, is4 in zip(gemms, self.four_wave) if is4 == want]
                if rs:
                    fl, ms = sum(f for f, _ in rs), sum(t for _, t in rs)
                    split[name] = dict(launches=len(rs), tflops=fl / ms / 1e9, avg_us=1e3 * ms / len(rs), ms=ms)
            out["gemm"]["by_kernel"] = split
        return out


def pmc_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (scripts/pmc_traffic.sh -> profiles/r4_pmc_traffic.json; FETCH_SIZE*2 + WRITE_SIZE, separate passes).
    Counters cannot be collected inside the timed run, so the bench line cites the committed measurement."""
    for name in PMC_FILES:
        try:
            return json.load(open(os.path.join(ROOT, "profiles", name)))[kernel]["hbm_bytes_per_launch"]
        except Exception:
            continue
    return None


def pmc_dram(kernel: str):
    """the part of `pmc_traffic(kernel)` that HBM (not the Infinity Cache) served, from the TCC_EA0_*_DRAM request counters of the same
    committed PMC file (round 6 on; None where the file has no such pass)"""
    for name in PMC_FILES:
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))[kernel]
        except Exception:
            continue
        return (d.get("dram") or {}).get("dram_bytes_per_launch_est")
    return None


def pmc_mfma_busy(kernel: str):
    """SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x GRBM_GUI_ACTIVE) of `kernel` from the same committed PMC file (None if it has no such row)"""
    for name in PMC_FILES:
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))[kernel]
        except Exception:
            continue
        return (d.get("mfma") or {}).get("mfma_pipe_utilisation")
    return None


def pmc_traffic_file(kernel: str):
    """the committed PMC file that holds `kernel`'s traffic (the newest one that has a row for it)"""
    for name in PMC_FILES:
        try:
            json.load(open(os.path.join(ROOT, "profiles", name)))[kernel]["hbm_bytes_per_launch"]
            return "profiles/" + name
        except Exception:
            continue
    return None


def gemm_traffic(by_kernel: dict):
    """launch-weighted HBM bytes per launch over the GEMM kernels of the step, each from ITS OWN PMC row; None if a kernel
    that served launches has no committed PMC row (a figure of another kernel is not evidence for it)"""
    tot, n = 0.0, 0
    for k, v in (by_kernel or {}).items():
        t = pmc_traffic(k)
        if t is None:
            return None
        tot += t * v["launches"]
        n += v["launches"]
    return tot / n if n else None


PMC_FILES = ("r6_pmc_traffic.json", "r5_pmc_traffic.json", "r4_pmc_traffic.json", "r3_pmc_traffic.json", "r2_pmc_traffic.json")


def pmc_source() -> str:
    for name in PMC_FILES:
        if os.path.exists(os.path.join(ROOT, "profiles", name)):
            return "profiles/" + name
    return "none"


def _tiled_state_dict(cfg: dict, seed: int = 0) -> dict:
    """fp32 weights for TIMING the CPU oracle: the value rule of oracle.synth_param (matrices ~ N(0, 1 / fan_in), damped AdaLN
    linears, norm weights near 1) with the normal draws of the matrices taken from ONE seeded 4 M-element block, tiled - drawing
    the 4.3 G normals of the full model costs ~90 s of host time, tiling them a few seconds, and a forward's run time does
    not depend on the values"""
    from oracle import ctsd_oracle as O
    gen = torch.Generator().manual_seed(seed)
    block = torch.randn(1 << 22, generator=gen)
    sd = {}
    for name, shape in O.param_shapes(cfg).items():
        n = 1
        for d in shape:
            n *= d
        if len(shape) < 2 or name == "pos_embed.pos_embed" or name.endswith("mix_factor") or n <= block.numel():
            sd[name] = O.synth_param(name, shape, cfg, gen)
            continue
        fan_in = n // shape[0]
        std = fan_in ** -0.5
        if ".norm1.linear" in name or ".norm1_context.linear" in name or name.startswith("norm_out.linear"):
            std *= 0.5
        out = torch.empty(n)
        k = n // block.numel()
        out[:k * block.numel()].view(k, block.numel()).copy_(block)
        out[k * block.numel():].copy_(block[:n - k * block.numel()])
        sd[name] = out.mul_(std).view(shape)
    return sd


def cpu_baseline(threads: int, layout: bool, step_flops: float):
    """The reference's CPU path (its fp32 PyTorch graph, restated in oracle/ctsd_oracle.py) on the host cores.
    `value` / `cores` are what THIS run measured (below), extrapolated to the step; the one whole denoise step ever timed on a GPU box's
    host (1526 s on 248 threads, round 2, another host, profiles/r2_cpu_full_step.json) is cited under `cited_full_step` only - a whole
    step does not fit a bench run.  Timed IN THIS RUN, as the bounded sample: full-depth CFG forwards of the step's model on 6 views x 1 frame and 6 views x 2 frames (1/16 and 2/16 of the images of the
    timed step; every layer, the full width, the full text length, the layout adapter when the step has it).  A step's FLOPs are linear
    in the number of images except for the temporal attention (0.7 %); the two samples show how far the host's time is
    (`sample_linearity` = t(2 frames) / (2 t(1 frame))); `value` is the 2-frame sample's extrapolation to 16 frames (it flatters the
    CPU: the small samples fit the host's caches better than the full step does)."""
    from oracle import ctsd_oracle as O
    from opendwm_amd.dit import model_flops
    torch.set_num_threads(threads)
    kwargs = variant_kwargs(layout)
    cfg = O.make_config(**kwargs)
    t0 = time.perf_counter()
    sd = _tiled_state_dict(cfg)
    t_weights = time.perf_counter() - t0
    w = WORKLOAD
    secs, finite = {}, True
    for frames in (1, 2):
        inp = O.make_inputs(cfg, 2, frames, w["V"], w["H"], w["W"], seed=0, text_len=w["text_len"], n_time_ids=13 if layout else 11)
        if layout:
            inp["condition_image_tensor"] = torch.rand(2, frames, w["V"], 6, 8 * w["H"], 8 * w["W"], generator=torch.Generator().manual_seed(77))
        with torch.no_grad():
            t0 = time.perf_counter()
            out = O.dit_forward(sd, cfg, **inp)
            secs[frames] = time.perf_counter() - t0
        finite = finite and bool(torch.isfinite(out).all())
        del out, inp
    fl = model_flops(kwargs, 2, 1, w["V"], w["H"], w["W"], w["text_len"])
    sample_flops = fl["total"] + (fl["adapter"] if layout else 0)
    extrap = {f"from_{f}_frame{'s' if f > 1 else ''}": 1.0 / (w["T"] / f * secs[f]) for f in secs}
    res = dict(value=extrap["from_2_frames"], unit="denoise-steps/s", cores=threads, kind="port",
               sample=f"measured in this run: full-depth CFG forwards of the fp32 PyTorch-CPU oracle "
                      f"({'text+layout' if layout else 'text-only'} model, {kwargs['num_layers']} layers, d = 1536, {w['text_len']} text tokens) "
                      f"on {w['V']} views x 1 frame ({secs[1]:.1f} s, {sample_flops / 1e12:.1f} TFLOP = {sample_flops / secs[1] / 1e12:.3f} TFLOP/s) and x 2 frames "
                      f"({secs[2]:.1f} s) x {w['H']}x{w['W']} latents = 1/{w['T']} and 2/{w['T']} of the step's images, on {threads} threads of "
                      f"{os.cpu_count()} host CPUs; weights tiled from a seeded block in {t_weights:.0f} s (untimed)",
               seconds_measured=secs[2], seconds_by_frames={str(k): v for k, v in secs.items()}, sample_linearity=secs[2] / (2.0 * secs[1]),
               extrapolated_from_samples=extrap, sample_flop=sample_flops, step_flop=step_flops, finite=finite)
    # `value` and `cores` describe what THIS run measured (the 2-frame sample on `threads` threads, extrapolated linearly to the
    # step's 16 frames).  The one full-size step ever timed (round 2, another host, 248 threads) is cited beside it, never as `value`,
    # and only when it is the same model variant as this run's.
    res["value_source"] = "extrapolated from this run's 2-frame sample (seconds_measured) on `cores` threads"
    try:
        m = json.load(open(os.path.join(ROOT, "profiles", "r2_cpu_full_step.json")))
        want = "text+layout" if layout else "text-only"
        cited = dict(value=m["denoise_steps_per_s"], seconds_per_step=m["seconds_per_step"], threads=m["threads"],
                     host_cores=m["host_cores"], variant=m["variant"], source="profiles/r2_cpu_full_step.json "
                     "(scripts/cpu_full_step.py: one FULL-SIZE step of the same oracle, timed once in round 2 on another host)")
        cited["same_variant_as_this_run"] = str(m["variant"]).startswith(want)
        res["cited_full_step"] = cited
    except Exception:
        pass
    return res


UNET_KWARGS = dict(   # examples/ctsd_21_6views_video_generation.json (reference repo)
    addition_time_embed_dim=256, block_out_channels=(320, 640, 1280, 1280), cross_attention_dim=1024,
    down_block_types=("CrossAttnDownBlockCrossviewTemporal",) * 3 + ("DownBlockCrossviewTemporal",), in_channels=4,
    layers_per_block=2, num_attention_heads=(5, 10, 20, 20), out_channels=4, projection_class_embeddings_input_dim=2816,
    sample_size=96, transformer_layers_per_block=1,
    up_block_types=("UpBlockCrossviewTemporal",) + ("CrossAttnUpBlockCrossviewTemporal",) * 3, enable_crossview=True,
    enable_rowwise_crossview=True, enable_temporal=True, enable_rowwise_temporal=True, merge_factor=2)
UNET_WORKLOAD = dict(B=1, T=6, V=6, C=4, H=32, W=56, text_len=77, guidance_scale=3.0, inference_steps=50)


def main_unet(args):
    """SD 2.1 UNet denoise step (BASELINE configs[1]): UNet forward at the CFG batch + guidance + DPM-Solver++ update."""
    from opendwm_amd import dist as D
    rank, local_rank, world = D.env_ranks()
    assert world == args.gpus, f"WORLD_SIZE {world} != --gpus {args.gpus}"
    dev = D.local_device(local_rank)          # (set_device included)
    D.init("nccl", dev)
    from opendwm_amd import _lib
    from opendwm_amd.pipeline import UNetDenoiser
    from opendwm_amd.unet import UNetCrossviewTemporalConditionModel, unet_flops
    from opendwm_amd.build import ensure_built
    ensure_built()                  # no-op when the in-tree libdwm_hip.so travelled with the snapshot
    _lib.load()
    timer = KernelTimer().install()
    import opendwm_amd.unet as unet_mod
    import opendwm_amd.ops as ops_mod
    unet_mod.ops = ops_mod
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device(dev):
            model = UNetCrossviewTemporalConditionModel(**UNET_KWARGS)
    finally:
        torch.set_default_dtype(old)
    model = model.to(device=dev, dtype=torch.bfloat16).eval()
    g = torch.Generator(device="cuda").manual_seed(0)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("mix_factor"):
                p.fill_(2.0)
            elif p.dim() == 1:
                p.normal_(0.0, 0.05, generator=g)
                if name.endswith(".weight"):
                    p.add_(1.0)
            else:
                std = p[0].numel() ** -0.5
                if ".conv2." in name or name.endswith("proj_out.weight") or ".net.2." in name or ".to_out.0." in name:
                    std *= 0.5
                p.normal_(0.0, std, generator=g)
    w = UNET_WORKLOAD
    gi = torch.Generator(device="cuda").manual_seed(1000 + rank)
    B2, T, V = 2 * w["B"], w["T"], w["V"]
    ring = torch.zeros(V, V, dtype=torch.bool)
    for i in range(V):
        for d in (-1, 0, 1):
            ring[i, (i + d) % V] = True
    cond = dict(
        encoder_hidden_states=(torch.randn(B2, T, V, w["text_len"], 1024, device=dev, generator=gi) * 0.5).to(torch.bfloat16),
        disable_crossview=torch.zeros(B2, dtype=torch.bool, device=dev), disable_temporal=torch.zeros(B2, dtype=torch.bool, device=dev),
        crossview_attention_mask=ring[None].repeat(B2, 1, 1).to(dev),
        added_time_ids=torch.rand(B2, T, V, 11, device=dev, generator=gi) * 2 - 1)
    latents = torch.randn(w["B"], T, V, w["C"], w["H"], w["W"], device=dev, generator=gi)
    den = UNetDenoiser(model, guidance_scale=w["guidance_scale"], inference_steps=w["inference_steps"]).prepare(latents, cond)
    ninf = w["inference_steps"]

    def step(i):
        timer.enabled = i >= args.warmup
        den.step(i % (ninf - 1))

    dt = D.timed_steps(step, args.steps, args.warmup, dev)
    timer.enabled = False
    finite = bool(torch.isfinite(den.latents).all().item())
    if rank == 0:
        fl = unet_flops(UNET_KWARGS, B2, T, V, w["H"], w["W"], w["text_len"])
        ks = timer.summary()
        if args.gemm_shapes:
            for row in timer.shape_table():
                print(json.dumps(row), file=sys.stderr)
        step_ms = 1e3 * dt / args.steps
        gm, at = ks.get("gemm", {}), ks.get("attn", {})
        print(json.dumps({
            "metric": "denoise-steps/sec (6-view x6f 448x256), SD-2.1 CTSD UNet", "value": world * args.steps / dt,
            "unit": "denoise-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic (seeded random-init weights, random latents / text embeddings)",
            "config": {"workload": "BASELINE configs[1]: CTSD SD-2.1 cross-view temporal UNet (row-wise cross-view + temporal blocks), "
                                   "6 views x 6 frames x 448x256 px (latents [1,6,6,4,32,56]), CFG g=3 -> model batch 2, 77 text tokens, "
                                   "DPM-Solver++(2M); one replica per GPU",
                       "flop_per_step": fl, "finite": finite, "parameters": sum(p.numel() for p in model.parameters())},
            "roofline": {"bound": "mfma", "kernel": "gemm_bf16_kernel (linear + implicit-GEMM conv, all epilogues)",
                         "achieved": gm.get("tflops"), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": (gm.get("tflops") or 0.0) / PEAK_BF16_TFLOPS, "traffic": None, "launches": gm.get("launches"),
                         "avg_launch_us": gm.get("avg_us"), "share_of_step_time": (gm.get("ms", 0.0) / args.steps) / step_ms},
            "roofline_attention": {"bound": "mfma", "kernel": "attn_res_kernel / attn_fwd_kernel (self / row-wise)", "achieved": at.get("tflops"),
                                   "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": (at.get("tflops") or 0.0) / PEAK_BF16_TFLOPS,
                                   "launches": at.get("launches"), "avg_launch_us": at.get("avg_us"),
                                   "share_of_step_time": (at.get("ms", 0.0) / args.steps) / step_ms},
            "whole_step_mfma_frac": fl / (step_ms * 1e-3) / (PEAK_BF16_TFLOPS * 1e12),
        }))
    D.shutdown()


def main_train(args):
    """Training step (ctsd.py:1195-1437, SD 3 branch) on synthetic latents / conditions: fp32 master weights, bf16
    compute, checkpointed blocks, HIP backward kernels, HIP AdamW; with N > 1 ranks torch DDP all-reduces the
    gradients over RCCL (one sample per GPU: weak scaling)."""
    from opendwm_amd import dist as D
    rank, local_rank, world = D.env_ranks()
    assert world == args.gpus, f"WORLD_SIZE {world} != --gpus {args.gpus}"
    dev = D.local_device(local_rank)          # (set_device included)
    D.init("nccl", dev, force=args.preflight)     # --preflight on one GPU: a one-rank RCCL group, DDP and its probes included
    pre = D.preflight(dev) if (world > 1 or args.preflight) else None
    use_ddp = world > 1 or args.preflight
    from opendwm_amd import _lib
    from opendwm_amd.dit import DiTCrossviewTemporalConditionModel, model_flops
    from opendwm_amd.pipeline import CTSDTrainer
    from opendwm_amd.build import ensure_built
    ensure_built()                  # no-op when the in-tree libdwm_hip.so travelled with the snapshot
    _lib.load()
    kwargs = dict(MODEL_KWARGS)
    if args.layers is not None:
        n = args.layers
        kwargs.update(num_layers=n, dual_attention_layers=[i for i in kwargs["dual_attention_layers"] if i < n],
                      crossview_block_layers=[i for i in kwargs["crossview_block_layers"] if i < n],
                      temporal_block_layers=[i for i in kwargs["temporal_block_layers"] if i < n])
    with torch.device(dev):
        model = DiTCrossviewTemporalConditionModel(**kwargs)          # fp32 master parameters
    synth_init_(model, 0)                                             # same seed on every rank
    if args.freeze_base:
        model.transformer_blocks.requires_grad_(False)
        model.time_text_embed.requires_grad_(False)
    n_train = sum(p.numel() for p in model.parameters() if p.requires_grad)
    n_all = sum(p.numel() for p in model.parameters())
    trainer = CTSDTrainer(model, lr=1e-5, weight_decay=0.01, ddp=use_ddp)
    w = WORKLOAD
    cond = {k: (v[:w["B"]] if torch.is_tensor(v) else v) for k, v in make_conditions(dev, seed=rank).items()}
    g = torch.Generator(device="cuda").manual_seed(rank)
    latents = torch.randn(w["B"], w["T"], w["V"], w["C"], w["H"], w["W"], device=dev, generator=g)
    gen = torch.Generator().manual_seed(1234 + rank)
    losses = []

    def step(i):
        losses.append(trainer.train_step(latents, cond, generator=gen))

    dt = D.timed_steps(step, args.steps, args.warmup, dev)
    finite = bool(torch.isfinite(torch.stack(losses)).all().item())
    # gradient exchange (ctsd.py:1051-1054: DDP all-reduce), outside the timed region: the same step without synchronisation
    # (DDP no_sync) and one all-reduce of the gradient bytes on its own -> how much of the exchange hides behind the backward
    # The probes run forward + backward only and drop the gradients (no optimizer step: the replicas stay identical), the same
    # number of times with and without synchronisation.
    ddp_extra = None
    if use_ddp:
        import contextlib

        def fwd_bwd(sync: bool):
            def run(i):
                with (contextlib.nullcontext() if sync else trainer.wrapper.no_sync()):
                    trainer.loss(latents, cond, generator=gen).backward()
                trainer.optimizer.zero_grad()
            return run
        n_probe = 2
        dt_sync = D.timed_steps(fwd_bwd(True), n_probe, 1, dev)
        dt_ns = D.timed_steps(fwd_bwd(False), n_probe, 1, dev)
        wire = torch.bfloat16 if getattr(trainer, "ddp_comm_dtype", None) == torch.bfloat16 else torch.float32
        nbytes = n_train * (2 if wire == torch.bfloat16 else 4)
        ar_ms = D.measure_allreduce(nbytes, dev, wire)
        exposed = max(0.0, 1e3 * (dt_sync - dt_ns) / n_probe)
        timeline = ddp_bucket_timeline(trainer, lambda: trainer.loss(latents, cond, generator=gen), dev, world)
        ddp_extra = dict(bucket_timeline=timeline, allreduce_ms=ar_ms, allreduce_bytes=nbytes, wire_dtype=str(wire).replace("torch.", ""),
                         forward_backward_ms_with_gradient_sync=1e3 * dt_sync / n_probe,
                         forward_backward_ms_without_gradient_sync=1e3 * dt_ns / n_probe, exposed_allreduce_ms=exposed,
                         backward_overlap_frac=(1.0 - min(1.0, exposed / ar_ms)) if ar_ms > 0 else None)
    if rank == 0:
        fwd = model_flops(kwargs, w["B"], w["T"], w["V"], w["H"], w["W"], w["text_len"])["total"]
        step_ms = 1e3 * dt / args.steps
        mem = torch.cuda.max_memory_allocated(dev) / 2 ** 30
        print(json.dumps({
            "metric": "train-samples/sec (6-view x16f 448x256 per sample), SD-3.5 CTSD train step", "value": world * args.steps / dt,
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16 compute, fp32 master weights / grads / AdamW",
            "data": "synthetic (seeded random-init weights, random latents / text embeddings)",
            "config": {"workload": "BASELINE config 4: CTSD SD-3.5 MMDiT train step (flow-matching loss, checkpointed blocks: "
                                   "forward + recompute + backward, AdamW), one [1,16,6,16,32,56] sample per GPU, DDP over RCCL",
                       "layers": kwargs["num_layers"], "parameters": n_all, "trainable_parameters": n_train,
                       "frozen_base": bool(args.freeze_base), "forward_flop": fwd,
                       "loss_first": float(losses[0]), "loss_last": float(losses[-1]), "finite": finite,
                       "peak_memory_GiB": mem,
                       # DDP: one bucketed all-reduce of the trainable gradients per step, bf16 on the wire (bf16_compress_hook,
                       # 200 MB buckets), overlapped with the rest of the backward; a ring over N GPUs moves 2 (N-1)/N of it per GPU
                       "ddp": ddp_extra, "preflight": pre},
            # forward + recompute + 2x backward GEMMs (input + weight gradients); frozen weights skip their wgrad
            "approx_mfma_frac": (4.0 * fwd) / (step_ms * 1e-3) / (PEAK_BF16_TFLOPS * 1e12) if not args.freeze_base else None,
        }))
    D.shutdown()


def ddp_bucket_timeline(trainer, loss_fn, dev, world: int, n_gpus_model: int = 8, link_GBps: float = 153.0):
    """When does the backward hand each DDP bucket to the gradient exchange, and how much of an N-GPU exchange would that leave exposed?
    One extra forward + backward with a comm hook that stamps an event on the backward's stream when a bucket is ready (then runs the
    default all-reduce).  From the stamps, for `n_gpus_model` GPUs on xGMI (no 8-GPU node is at hand: a model, stated as one): buckets go
    out in ready order, one at a time, a ring all-reduce of b bytes moves 2 (N-1)/N b through every GPU over two links (one per
    neighbour and direction, `link_GBps` each: DESIGN.md section 5), a direct reduce-scatter + all-gather over all N-1 links the same bytes
    over N-1 links; exposed = what is still running when the backward ends.  fp32 and bf16 wire."""
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
    recs = []

    def hook(state, bucket):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream(dev))
        recs.append((bucket.index(), bucket.buffer().numel(), ev))
        return default_hooks.allreduce_hook(state, bucket)

    try:
        trainer.wrapper.register_comm_hook(None, hook)
    except Exception as e:          # a hook is already registered (bf16 wire): no second one
        return {"error": f"no timeline: {e!r}"}
    t0, t1, t2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    for it in range(2):             # the second pass is the one reported (the first rebuilds the buckets in ready order)
        recs.clear()
        t0.record(torch.cuda.current_stream(dev))
        loss = loss_fn()
        t1.record(torch.cuda.current_stream(dev))
        loss.backward()
        t2.record(torch.cuda.current_stream(dev))
        torch.cuda.synchronize(dev)
        trainer.optimizer.zero_grad()
    bwd_ms = t1.elapsed_time(t2)
    ready = sorted((t1.elapsed_time(ev), n) for _, n, ev in recs)
    N = n_gpus_model

    def exposed(bytes_per_elem, links):
        end = 0.0
        for t, n in ready:
            ms = 2.0 * (N - 1) / N * n * bytes_per_elem / (links * link_GBps * 1e9) * 1e3
            end = max(end, t) + ms
        return max(0.0, end - bwd_ms), sum(2.0 * (N - 1) / N * n * bytes_per_elem / (links * link_GBps * 1e9) * 1e3 for _, n in ready)

    out = {"what": f"bucket-ready stamps of one backward on this GPU; exchange times MODELLED for {N} GPUs (xGMI {link_GBps:.0f} GB/s per link and direction), not measured",
           "forward_ms": t0.elapsed_time(t1), "backward_ms": bwd_ms, "buckets": len(ready), "elements": int(sum(n for _, n in ready)),
           "ready_ms_after_backward_start": [round(t, 1) for t, _ in ready],
           "backward_left_after_ready_ms": [round(bwd_ms - t, 1) for t, _ in ready]}
    for name, bpe in (("fp32", 4), ("bf16", 2)):
        for alg, links in (("ring_2_links", 2), ("direct_rs_ag_7_links", N - 1)):
            ex, tot = exposed(bpe, links)
            out[f"{name}_{alg}"] = {"exchange_ms": round(tot, 1), "exposed_ms": round(ex, 1)}
    return out


def _unet_synth_init_(model, seed: int):
    """the UNet's seeded synthetic initialisation (bench only)"""
    g = torch.Generator(device="cuda").manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("mix_factor"):
                p.fill_(2.0)
            elif p.dim() == 1:
                p.normal_(0.0, 0.05, generator=g)
                if name.endswith(".weight"):
                    p.add_(1.0)
            else:
                std = p[0].numel() ** -0.5
                if ".conv2." in name or name.endswith("proj_out.weight") or ".net.2." in name or ".to_out.0." in name:
                    std *= 0.5
                p.normal_(0.0, std, generator=g)


def main_train_unet(args):
    """SD 2.1 training step (ctsd.py:1195-1437 in its UNet branch :1240-1253) at the BASELINE configs[1] geometry: DDPM noising,
    v-prediction target, UNet forward + recompute + backward through opendwm_amd.train_unet, HIP AdamW on fp32 masters."""
    from opendwm_amd import dist as D
    rank, local_rank, world = D.env_ranks()
    assert world == args.gpus, f"WORLD_SIZE {world} != --gpus {args.gpus}"
    dev = D.local_device(local_rank)          # (set_device included)
    D.init("nccl", dev)
    from opendwm_amd import _lib
    from opendwm_amd.pipeline import CTSDTrainer
    from opendwm_amd.unet import UNetCrossviewTemporalConditionModel, unet_flops
    from opendwm_amd.build import ensure_built
    ensure_built()
    _lib.load()
    with torch.device(dev):
        model = UNetCrossviewTemporalConditionModel(**UNET_KWARGS)          # fp32 master parameters
    _unet_synth_init_(model, 0)                                            # same seed on every rank
    n_all = sum(p.numel() for p in model.parameters())
    trainer = CTSDTrainer(model, lr=1e-5, weight_decay=0.01, ddp=world > 1)
    w = UNET_WORKLOAD
    B, T, V = w["B"], w["T"], w["V"]
    gi = torch.Generator(device="cuda").manual_seed(1000 + rank)
    ring = torch.zeros(V, V, dtype=torch.bool)
    for i in range(V):
        for d in (-1, 0, 1):
            ring[i, (i + d) % V] = True
    cond = dict(
        encoder_hidden_states=(torch.randn(B, T, V, w["text_len"], 1024, device=dev, generator=gi) * 0.5).to(torch.bfloat16),
        disable_crossview=torch.zeros(B, dtype=torch.bool, device=dev), disable_temporal=torch.zeros(B, dtype=torch.bool, device=dev),
        crossview_attention_mask=ring[None].repeat(B, 1, 1).to(dev),
        added_time_ids=torch.rand(B, T, V, 11, device=dev, generator=gi) * 2 - 1)
    latents = torch.randn(B, T, V, w["C"], w["H"], w["W"], device=dev, generator=gi)
    gen = torch.Generator().manual_seed(1234 + rank)
    losses = []

    def step(i):
        losses.append(trainer.train_step(latents, cond, generator=gen))

    dt = D.timed_steps(step, args.steps, args.warmup, dev)
    finite = bool(torch.isfinite(torch.stack(losses)).all().item())
    if rank == 0:
        fwd = unet_flops(UNET_KWARGS, B, T, V, w["H"], w["W"], w["text_len"])
        step_ms = 1e3 * dt / args.steps
        print(json.dumps({
            "metric": "train-samples/sec (6-view x6f 448x256 per sample), SD-2.1 CTSD UNet train step", "value": world * args.steps / dt,
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16 compute, fp32 master weights / grads / AdamW",
            "data": "synthetic (seeded random-init weights, random latents / text embeddings)",
            "config": {"workload": "CTSD SD-2.1 UNet train step (DDPM noising, v-prediction loss, checkpointed blocks: forward + "
                                   "recompute + backward, AdamW), one [1,6,6,4,32,56] sample per GPU, DDP over RCCL",
                       "parameters": n_all, "forward_flop": fwd, "loss_first": float(losses[0]), "loss_last": float(losses[-1]),
                       "finite": finite, "peak_memory_GiB": torch.cuda.max_memory_allocated(dev) / 2 ** 30},
            "approx_mfma_frac": (4.0 * fwd) / (step_ms * 1e-3) / (PEAK_BF16_TFLOPS * 1e12),
        }))
    D.shutdown()


# BASELINE configs[4]: examples/ctsd_35_tvae_6views_video_generation_with_layout.json (reference repo) - inference_config :55-69
TVAE_AR = dict(frames=40, sequence_length_per_iteration=17, vae_pre=1, vae_stride=4, reference_frame_count=1, guidance_scale=4.0,
               inference_steps=40, memory_efficient_batch=2, n_time_ids=11)


def main_tvae_ar(args):
    """BASELINE configs[4] on one GPU: the SD 3.5 CTSD text+layout model over the CogVideoX temporal VAE, 6 views x 40 frames
    generated autoregressively as the reference does (ctsd.py:1656-1833 with the inference_config of the example JSON): windows
    of 17 frames = 5 latent frames (vae_pre 1, vae_stride 4), stride 16, the last latent frame of a window is the clean
    reference frame of the next one; every window = 40 guided FlowMatch-Euler steps of the FULL-SIZE model on latents
    [1,5,6,16,32,56] + one temporal-VAE decode of 6 clips x 17 frames x 256x448 (memory_efficient_batch 2).  40 frames -> 2
    windows -> 33 generated frames (the reference's `range(0, total - length + 1, stride)` drops the ragged tail).  The timed
    region is the whole job: both windows' denoise loops and both decodes; conditions (text / layout tensors) are resident, as
    for the headline.  --steps = inference steps per window (default: the example's 40), --warmup = untimed steps of a
    warm-up window (and one untimed decode)."""
    from opendwm_amd import dist as D
    rank, local_rank, world = D.env_ranks()
    assert world == args.gpus == 1, "--tvae-ar is a one-GPU line (the multi-GPU form of this job is --frame-shard / replicas)"
    dev = D.local_device(local_rank)          # (set_device included)
    from opendwm_amd import _lib
    from opendwm_amd.build import ensure_built
    from opendwm_amd.dit import model_flops
    from opendwm_amd.drivers import AutoregressiveDriver, LatentDecoder, LatentEncoder, latent_sequence_length
    from opendwm_amd.pipeline import CTSDDenoiser
    from opendwm_amd.vae_cogvideox import AutoencoderKLCogVideoX
    ensure_built()
    _lib.load()
    c, w = TVAE_AR, WORKLOAD
    steps = args.steps if args.steps_given else c["inference_steps"]
    kwargs = variant_kwargs(True)
    kwargs.update(projection_class_embeddings_input_dim=256 * c["n_time_ids"])           # "fps_camera_transforms": 11 ids
    timer = KernelTimer().install()
    try:
        model = build_model(kwargs, dev, seed=0)
        if args.residual_bf16:
            model.residual_dtype = torch.bfloat16
        model.cache_adapter_residuals = bool(args.adapter_cache)
        vae = AutoencoderKLCogVideoX().to(dev).to(torch.bfloat16).eval()                  # THUDM/CogVideoX-2b widths
        synth_init_(vae, 1)
        T_lat = latent_sequence_length(c["sequence_length_per_iteration"], c["vae_pre"], c["vae_stride"])
        wl = dict(w, T=T_lat)
        icfg = dict(inference_steps=steps, sequence_length_per_iteration=c["sequence_length_per_iteration"], vae_pre=c["vae_pre"],
                    vae_stride=c["vae_stride"], reference_frame_count=c["reference_frame_count"],
                    autoregression_data_exception_for_take_sequence=["crossview_mask"])
        den = CTSDDenoiser(model, guidance_scale=c["guidance_scale"], inference_steps=steps)
        if args.graph:
            den.enable_graph()
        dec_events = []

        class TimedDecoder(LatentDecoder):
            def __call__(self, latents, diffusion_forcing=False):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                out = super().__call__(latents, diffusion_forcing)
                e.record()
                dec_events.append((s, e))
                return out
        decode = TimedDecoder(vae, memory_efficient_batch=c["memory_efficient_batch"])
        drv = AutoregressiveDriver(den, icfg, decode=decode, generator=torch.Generator().manual_seed(0))
        plan = drv.plan(T_lat, c["frames"], False)
        # the windows' conditions (what get_conditions hands the model: CFG-doubled text / pooled / time ids / layout images per
        # LATENT frame), built before the timed region
        conds = {wd.clip: make_conditions(dev, seed=wd.clip[0], w=wl, n_time_ids=c["n_time_ids"], layout=True) for wd in plan}
        shape = (w["B"], T_lat, w["V"], w["C"], w["H"], w["W"])
        # warm-up: a short window + one decode (kernels, allocator, clocks)
        g = torch.Generator(device="cuda").manual_seed(1)
        lat0 = torch.randn(shape, device=dev, generator=g)
        den.prepare(lat0, conds[plan[0].clip])
        for i in range(max(1, args.warmup)):
            den.step(i % steps)
        decode(den.result())
        dec_events.clear()
        timer.enabled = not args.graph
        D.sync(dev)
        t0 = time.perf_counter()
        out = drv.run(shape, lambda a, b: conds[(a, b)], c["frames"], dev)
        D.sync(dev)
        dt = time.perf_counter() - t0
        timer.enabled = False
        img = out["images"]
        finite = bool(torch.isfinite(img).all().item())
        decode_ms = sum(s.elapsed_time(e) for s, e in dec_events)
        # outside the timed region: the reference-frame encode an autoregressive CONTINUATION starts from
        # (generate_frames_for_reference = false, ctsd.py:1677-1703): 6 views x 1 frame -> 1 latent frame
        px = torch.rand(w["B"], 1, w["V"], 3, 8 * w["H"], 8 * w["W"], device=dev) * 2 - 1
        enc = LatentEncoder(vae, memory_efficient_batch=c["memory_efficient_batch"])
        enc(px, sample=False)
        torch.cuda.synchronize()
        te = time.perf_counter()
        ref_lat = enc(px, sample=False)
        torch.cuda.synchronize()
        enc_ms = 1e3 * (time.perf_counter() - te)
    finally:
        timer.uninstall()
    n_steps = steps * len(plan)
    frames = img.shape[0] // (w["B"] * w["V"])
    fl = model_flops(kwargs, 2 * w["B"], T_lat, w["V"], w["H"], w["W"], w["text_len"])
    step_flop = fl["total"] + (fl["adapter"] if not args.adapter_cache else 0)
    ks = timer.summary()
    gm, at = ks.get("gemm", {}), ks.get("attn", {})
    print(json.dumps({
        "metric": "denoise-steps/sec (6-view x40f as 17-frame windows over the CogVideoX tVAE), SD-3.5 CTSD, whole job incl. decode",
        "value": n_steps / dt, "unit": "denoise-steps/s", "n_gpus": 1, "steps": n_steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / n_steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
        "data": "synthetic (seeded random-init weights of the SD 3.5 CTSD model and of the CogVideoX-2b VAE architecture, random text / layout conditions)",
        "frames_per_s": frames / dt, "view_frames_per_s": img.shape[0] / dt, "seconds_total": dt,
        "seconds_denoise": dt - 1e-3 * decode_ms, "seconds_decode": 1e-3 * decode_ms, "reference_frame_encode_ms_untimed": enc_ms,
        "config": {"workload": "BASELINE.json configs[4] on one GPU: CTSD SD-3.5 MMDiT text+layout (24 layers, point-wise temporal, ImageAdapter) + "
                               "AutoencoderKLCogVideoX, 6 views x 40 frames 448x256 as autoregressive 17-frame windows (5 latent frames, stride 16, "
                               "1 reference frame), 40 guided FlowMatch-Euler steps per window, split decode (memory_efficient_batch 2)",
                   "windows": len(plan), "latent_window": list(shape), "frames_generated": frames, "images": list(img.shape),
                   "inference_steps_per_window": steps, "flop_per_step": step_flop, "finite": finite, "hip_graph": bool(args.graph),
                   "reference_latent": list(ref_lat.shape),
                   "residual_stream": "bf16" if args.residual_bf16 else "fp32 (GEMM operands and activations bf16)",
                   "baseline_config": "BASELINE.json configs[4] (examples/ctsd_35_tvae_6views_video_generation_with_layout.json)"},
        "roofline": None if not gm else {
            "bound": "mfma", "kernel": "gemm_bf16_kernel (all launches of the job: model GEMMs + the VAE's implicit-GEMM convolutions)",
            "achieved": gm.get("tflops"), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": (gm.get("tflops") or 0.0) / PEAK_BF16_TFLOPS,
            "traffic": None, "launches": gm.get("launches"), "avg_launch_us": gm.get("avg_us"),
            "share_of_job_time": gm.get("ms", 0.0) / (1e3 * dt)},
        "roofline_attention": None if not at else {"bound": "mfma", "kernel": "attn_res_kernel (joint L=602, dual L=448)", "achieved": at.get("tflops"),
                                                   "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": (at.get("tflops") or 0.0) / PEAK_BF16_TFLOPS},
        "denoise_mfma_frac": step_flop * n_steps / max(dt - 1e-3 * decode_ms, 1e-9) / (PEAK_BF16_TFLOPS * 1e12),
    }))


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` started plainly (no WORLD_SIZE in the environment): re-execute this command line
    under torch.distributed.run with one rank per GPU (one process per GPU as src/dwm/train.py:60-67 of the reference
    is launched); rank 0 of that job prints the JSON line."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL across processes needs it on this driver)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main_debug_cpu(args):
    """--debug-cpu-launch: the launch / timing path of this file on CPU (gloo) with a stand-in step - rank environment,
    process-group init, barrier-bracketed timing, MAX over ranks, one JSON line from rank 0.  No kernel runs: the line is
    NOT a measurement and says so."""
    from opendwm_amd import dist as D
    rank, local_rank, world = D.env_ranks()
    assert world == args.gpus, f"WORLD_SIZE {world} != --gpus {args.gpus}"
    D.init("gloo")
    pre = D.preflight(None, mbytes=1)            # the same preflight and gradient-exchange probes the GPU paths run
    ar_ms = D.measure_allreduce(1 << 20, None)
    x = torch.randn(64, 64)

    def step(i):
        (x @ x).sum().item()
        if rank == world - 1:
            time.sleep(0.02)             # the slowest rank defines the reported time

    dt = D.timed_steps(step, args.steps, args.warmup)
    if rank == 0:
        print(json.dumps({"metric": "DEBUG launch-path check (no kernels) - INVALID as a bench line", "value": world * args.steps / dt,
                          "unit": "stand-in steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "none", "config": {"workload": "debug-cpu-launch"},
                          "preflight": pre, "allreduce_ms": ar_ms}))
    D.shutdown()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stack-modulation", action="store_true",
                    help="A/B: the AdaLN modulation rows of every joint block from its own two M = images launches instead of one stacked GEMM "
                         "per step (model.stack_modulation, default on)")
    ap.add_argument("--preflight", action="store_true",
                    help="run the launch preflight (device / RCCL facts, checked all-reduce) also with one rank; always on for N > 1")
    ap.add_argument("--gemm-shapes", action="store_true", help="diagnostics: per-shape GEMM totals on stderr")
    ap.add_argument("--cfg-split", action="store_true",
                    help="pairs of ranks share one sample: CFG halves on two GPUs, one all-gather per step (per-sample latency "
                         "mode; the default is one replica per GPU)")
    ap.add_argument("--frame-shard", action="store_true",
                    help="all ranks share ONE sample, T/N frames each (opendwm_amd.sharding: an all-to-all before and after "
                         "every temporal block); per-sample latency mode, reported as strong scaling.  With --cfg-split: "
                         "2 CFG halves x N/2 frame shards")
    ap.add_argument("--graph", action="store_true",
                    help="replay the whole step as one HIP graph (CTSDDenoiser.enable_graph); the per-kernel HIP-event "
                         "roofline cannot be taken inside a graph, so the roofline fields are empty in this mode")
    ap.add_argument("--layers", type=int, default=None, help="debug: truncate the model (INVALID as a bench line)")
    ap.add_argument("--text-only", action="store_true",
                    help="the text-conditioned variant (examples/ctsd_35_6views_video_generation.json: row-wise temporal "
                         "attention, no ImageAdapter) instead of the default text+layout variant BASELINE.json configs[2] names "
                         "(examples/ctsd_35_df16_6views_video_generation_with_layout.json model: ImageAdapter + point-wise "
                         "temporal attention, 13 added time ids)")
    ap.add_argument("--layout", action="store_true", help="(default; kept for old command lines)")
    ap.add_argument("--no-text-only-leg", action="store_true",
                    help="skip the secondary leg that times the text-only variant after the headline (reported as 'text_only')")
    ap.add_argument("--residual-bf16", action="store_true",
                    help="A/B: keep the hidden / context streams in bf16 (model.residual_dtype; default: fp32 streams, the "
                         "setting the parity tests hold to the oracle)")
    ap.add_argument("--adapter-cache", action="store_true",
                    help="keep the ImageAdapter residuals across denoise steps (its input does not change from step to step); "
                         "the default recomputes the adapter inside every step as the reference's forward does")
    ap.add_argument("--train", action="store_true",
                    help="BASELINE config 4 instead of the headline metric: one SD-3.5 training step per 'step' "
                         "(forward + backward + AdamW on one 6-view x 16-frame sample per GPU, DDP gradient all-reduce over RCCL)")
    ap.add_argument("--unet", action="store_true",
                    help="BASELINE config 2 instead of the headline metric: SD-2.1 cross-view temporal UNet, 6 views x 6 frames "
                         "(examples/ctsd_21_6views_video_generation.json: DPM-Solver++ 50 steps, guidance 3)")
    ap.add_argument("--tvae-ar", action="store_true",
                    help="BASELINE config 5 on one GPU instead of the headline metric: 6 views x 40 frames as autoregressive 17-frame "
                         "windows over the CogVideoX temporal VAE (examples/ctsd_35_tvae_6views_video_generation_with_layout.json); "
                         "--steps = inference steps per window (default 40)")
    ap.add_argument("--freeze-base", action="store_true",
                    help="with --train: freezing_pattern ^(transformer_blocks|time_text_embed)$ of the reference's warm-up configs")
    ap.add_argument("--debug-cpu-launch", action="store_true",
                    help="debug: exercise the launch / timing path on CPU over gloo with a stand-in step (INVALID as a bench line)")
    args = ap.parse_args()
    args.steps_given = any(a == "--steps" or a.startswith("--steps=") for a in sys.argv[1:])
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    if args.debug_cpu_launch:
        return main_debug_cpu(args)
    if args.train and args.unet:
        return main_train_unet(args)
    if args.train:
        return main_train(args)
    if args.unet:
        return main_unet(args)
    if args.tvae_ar:
        return main_tvae_ar(args)

    from opendwm_amd import dist as D
    rank, local_rank, world = D.env_ranks()
    assert world == args.gpus, f"WORLD_SIZE {world} != --gpus {args.gpus}"
    dev = D.local_device(local_rank)          # (set_device included)
    D.init("nccl", dev, force=args.preflight)     # "nccl" == RCCL on ROCm; a single process needs no group unless --preflight asks for one
    pre = D.preflight(dev) if (world > 1 or args.preflight) else None     # fails loudly (rc != 0) before anything is timed

    from opendwm_amd import _lib
    from opendwm_amd.dit import model_flops
    from opendwm_amd.pipeline import CTSDDenoiser
    from opendwm_amd.build import ensure_built
    ensure_built()                  # no-op when the in-tree libdwm_hip.so travelled with the snapshot
    _lib.load()

    args.layout = not args.text_only
    w = WORKLOAD
    # --cfg-split: ranks (2k, 2k+1) share ONE sample (unconditional / conditional half each, one all-gather of the
    # prediction per step): N GPUs = N/2 samples in flight at about twice the per-sample speed
    cfg_group, sample_id, n_samples = None, rank, world
    if args.cfg_split:
        import torch.distributed as dist
        assert world % 2 == 0, "--cfg-split needs an even number of ranks"
        groups = [dist.new_group([2 * k, 2 * k + 1]) for k in range(world // 2)]
        cfg_group, sample_id, n_samples = groups[rank // 2], rank // 2, world // 2
    frame_group = None
    if args.frame_shard and world > 1:
        import torch.distributed as dist
        if args.cfg_split:              # ranks of equal parity hold the same CFG half: they split the frames
            fgroups = [dist.new_group(list(range(c, world, 2))) for c in range(2)]
            frame_group = fgroups[rank % 2]
        else:
            frame_group = dist.group.WORLD
        sample_id, n_samples = 0, 1
    ninf = w["inference_steps"]

    def run_variant(layout: bool, steps: int, warmup: int, adapter_cache: bool = bool(args.adapter_cache)):
        """build the model of one variant, W untimed + K timed denoise steps; -> (kwargs, seconds, KernelTimer, finite)"""
        kwargs = variant_kwargs(layout)
        if args.layers is not None:
            n = args.layers
            kwargs.update(num_layers=n, dual_attention_layers=[i for i in kwargs["dual_attention_layers"] if i < n],
                          crossview_block_layers=[i for i in kwargs["crossview_block_layers"] if i < n],
                          temporal_block_layers=[i for i in kwargs["temporal_block_layers"] if i < n])
        timer = KernelTimer().install()
        try:
            model = build_model(kwargs, dev, seed=0)
            if args.residual_bf16:
                model.residual_dtype = torch.bfloat16
            if args.no_stack_modulation:
                model.stack_modulation = False
            cond = make_conditions(dev, seed=sample_id, layout=layout)
            g = torch.Generator(device="cuda").manual_seed(sample_id)
            latents = torch.randn(w["B"], w["T"], w["V"], w["C"], w["H"], w["W"], device=dev, generator=g)
            den = CTSDDenoiser(model, guidance_scale=w["guidance_scale"], inference_steps=ninf,
                               cfg_group=cfg_group, frame_group=frame_group).prepare(latents, cond)
            if args.graph:
                den.enable_graph()

            model.cache_adapter_residuals = adapter_cache     # default: the adapter runs inside every timed step

            def step(i):
                timer.enabled = i >= warmup and not args.graph
                den.step(i % ninf)

            sampler = ClockPowerSampler(dev.index or 0).start() if rank == 0 else None
            dt = D.timed_steps(step, steps, warmup, dev)
            t_end = time.perf_counter()
            timer.enabled = False
            if sampler is not None:
                sampler.stop()
                timer.clock_power = sampler.summary(t_end - dt, t_end)
            finite = bool(torch.isfinite(den.latents).all().item())
        finally:
            timer.uninstall()
        del den, model, cond, latents
        torch.cuda.empty_cache()
        return kwargs, dt, timer, finite

    kwargs, dt, timer, finite = run_variant(args.layout, args.steps, args.warmup)
    # the other example of BASELINE config 3 (examples/ctsd_35_6views_video_generation.json, text only,
    # row-wise temporal attention, 398.98 TFLOP/step) - reported beside the headline, outside its timed region, at N=1 only
    other = None
    if world == 1 and args.layout and not args.no_text_only_leg and not args.graph and args.layers is None:
        try:
            k2, dt2, t2, fin2 = run_variant(False, args.steps, args.warmup)
            other = (k2, dt2, t2, fin2)
        except Exception as e:          # never lose the headline line to the secondary leg
            print(f"text-only leg failed: {e!r}", file=sys.stderr)

    # and the headline model with the layout residuals computed ONCE per prepare() (through the fp32 path) instead of inside
    # every step: the model class's own default (`cache_adapter_residuals`), what a deployment would run - reported beside the
    # headline, outside its timed region, at N=1 only
    cached = None
    if world == 1 and args.layout and not args.adapter_cache and not args.no_text_only_leg and not args.graph and args.layers is None:
        try:
            cached = run_variant(True, args.steps, args.warmup, adapter_cache=True)
        except Exception as e:
            print(f"adapter-cache leg failed: {e!r}", file=sys.stderr)

    if rank == 0:
        fl = model_flops(kwargs, 2 * w["B"], w["T"], w["V"], w["H"], w["W"], w["text_len"])
        step_flop = fl["total"] + (fl["adapter"] if args.layout and not args.adapter_cache else 0)
        ks = timer.summary()
        if args.gemm_shapes:
            for row in timer.shape_table():
                print(json.dumps(row), file=sys.stderr)
        step_ms = 1e3 * dt / args.steps
        gm, at, asm, agr = ks.get("gemm", {}), ks.get("attn", {}), ks.get("attn_small", {}), ks.get("attn_group", {})
        line = {
            "metric": "denoise-steps/sec (6-view x16f 448x256), SD-3.5 CTSD",
            "value": n_samples * args.steps / dt, "unit": "denoise-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_ms,
            "higher_is_better": True, "scaling": "strong" if frame_group is not None else "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic (seeded random-init weights, random latents / text embeddings)",
            "config": {"workload": "CTSD SD-3.5 MMDiT (24 joint blocks, 13 dual, 6 cross-view + 12 temporal VT blocks, "
                                   "" + ("point-wise temporal + ImageAdapter" if args.layout else "row-wise temporal") + "), 6 views x 16 frames x 448x256 px (latents [1,16,6,16,32,56]), CFG g=4 -> "
                                   "model batch 2, 154 text tokens, FlowMatch-Euler; " +
                                   ("one sample over all GPUs: " + ("2 CFG halves x " if args.cfg_split else "") + "frame shards, all-to-all around temporal blocks"
                                    if frame_group is not None else
                                    "CFG halves of one sample on two GPUs" if args.cfg_split else "one replica per GPU"),
                       "layers": kwargs["num_layers"], "flop_per_step": step_flop, "flop_model": fl["total"], "flop_adapter": fl["adapter"],
                       "baseline_config": "BASELINE.json configs[2]", "finite": finite,
                       "residual_stream": "bf16" if args.residual_bf16 else "fp32 (GEMM operands and activations bf16)",
                       "gemm_kernels": "4-wave (gemm_bf16_4w.hip: fast + general form) where they cover the launch, 8-wave otherwise (split-K launches, "
                                       "one K step, output row maps); DWM_GEMM4W=0 forces 8-wave",
                       "variant": ("text+layout (ImageAdapter recomputed every step, pointwise temporal)" if not args.adapter_cache else
                                   "text+layout (ImageAdapter residuals cached across steps, pointwise temporal)")
                       if args.layout else "text only (rowwise temporal, no adapter)"},
            "roofline": {"bound": "mfma", "kernel": "every dwm_gemm_bf16 launch of the step: gemm4w_kernel (gemm_bf16_4w.hip) where it covers the "
                                                    "launch, gemm_bf16_kernel otherwise (all epilogues; split in by_kernel)",
                         "achieved": gm.get("tflops"), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": (gm.get("tflops") or 0.0) / PEAK_BF16_TFLOPS, "traffic": gemm_traffic(gm.get("by_kernel")),
                         "traffic_unit": "L2 -> fabric bytes per launch (TCC_EA request counters: FETCH_SIZE x 2 + WRITE_SIZE in separate rocprofv3 PMC "
                                         "passes of this command, scripts/pmc_traffic.sh; each kernel's own row, launch-weighted over by_kernel).  "
                                         "Infinity-Cache (MALL) hits are counted, so this is an UPPER bound of the HBM bytes",
                         "by_kernel": {k: {kk: vv for kk, vv in v.items() if kk != "ms"} |
                                       {"share_of_step_time": (v["ms"] / args.steps) / step_ms, "traffic": pmc_traffic(k),
                                        "traffic_dram_est": pmc_dram(k), "traffic_source": pmc_traffic_file(k)}
                                       for k, v in (gm.get("by_kernel") or {}).items()},
                         "algorithmic_flop_per_launch": (gm.get("flops") or 0.0) / max(gm.get("launches") or 1, 1),
                         "launches": gm.get("launches"), "avg_launch_us": gm.get("avg_us"),
                         "share_of_step_time": (gm.get("ms", 0.0) / args.steps) / step_ms},
            "roofline_attention": {"bound": "mfma", "kernel": "attn_stream_kernel (attention_stream.hip: one wave per SIMD, V of a head double-buffered in LDS, K fragments "
                                                              "from global memory, Q in AGPRs with the softmax scale folded in by the q RMSNorm; joint L=602, dual L=448, "
                                                              "row-wise temporal L=448; round 5: attn_res_kernel)",
                                   "achieved": at.get("tflops"), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                   "frac": (at.get("tflops") or 0.0) / PEAK_BF16_TFLOPS,
                                   "launches": at.get("launches"), "avg_launch_us": at.get("avg_us"),
                                   "share_of_step_time": (at.get("ms", 0.0) / args.steps) / step_ms,
                                   "traffic": pmc_traffic("attn_stream_kernel"), "traffic_source": pmc_traffic_file("attn_stream_kernel"),
                                   "mfma_pipe_busy": pmc_mfma_busy("attn_stream_kernel"),
                                   "all_attention_launches_tflops": ((at.get("flops") or 0.0) + (asm.get("flops") or 0.0) + (agr.get("flops") or 0.0)) /
                                   max((at.get("ms") or 0.0) + (asm.get("ms") or 0.0) + (agr.get("ms") or 0.0), 1e-9) / 1e9},
            "roofline_attention_crossview": None if not agr else {
                "bound": "hbm", "kernel": "attn_group_lds_kernel (row-wise cross-view attention, ring view mask: one workgroup per (problem, head "
                                          "group), one wave per query view, K / V of a head copied to LDS once, allowed key views only)",
                "achieved": timer.group_bytes / (agr["ms"] * 1e-3) / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                "frac": timer.group_bytes / (agr["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                "algorithmic_bytes_per_launch": timer.group_bytes / agr["launches"], "launches": agr["launches"],
                "avg_launch_us": agr["avg_us"], "nominal_tflops": agr["tflops"], "share_of_step_time": (agr["ms"] / args.steps) / step_ms},
            "roofline_attention_pointwise": None if not asm else {
                "bound": "hbm", "kernel": "attn_small_kernel (point-wise temporal attention, L = frames)",
                "achieved": timer.small_bytes / (asm["ms"] * 1e-3) / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                "frac": timer.small_bytes / (asm["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                "algorithmic_bytes_per_launch": timer.small_bytes / asm["launches"], "launches": asm["launches"],
                "avg_launch_us": asm["avg_us"], "share_of_step_time": (asm["ms"] / args.steps) / step_ms},
            "whole_step_mfma_frac": step_flop / (step_ms * 1e-3) / (PEAK_BF16_TFLOPS * 1e12),
            "clock_power": getattr(timer, "clock_power", None),
        }
        cp = line["clock_power"]
        if cp and cp.get("peak_at_sustained_clock_tflops") and gm.get("tflops"):
            # the same launches against what the matrix pipes deliver at the clock the board sustained during this run
            line["roofline"]["frac_of_peak_at_sustained_clock"] = gm["tflops"] / cp["peak_at_sustained_clock_tflops"]
            if at.get("tflops"):
                line["roofline_attention"]["frac_of_peak_at_sustained_clock"] = at["tflops"] / cp["peak_at_sustained_clock_tflops"]
        if pre is not None:
            line["preflight"] = pre
        if other is not None:
            k2, dt2, t2, fin2 = other
            fl2, ks2 = model_flops(k2, 2 * w["B"], w["T"], w["V"], w["H"], w["W"], w["text_len"]), t2.summary()
            ms2 = 1e3 * dt2 / args.steps
            line["text_only"] = {
                "variant": "text only (examples/ctsd_35_6views_video_generation.json: row-wise temporal, no adapter), same "
                           "latents / CFG / scheduler, timed after the headline with the same --steps / --warmup",
                "value": n_samples * args.steps / dt2, "unit": "denoise-steps/s", "ms_per_step": ms2,
                "flop_per_step": fl2["total"], "finite": fin2,
                "gemm_tflops": ks2.get("gemm", {}).get("tflops"), "attention_tflops": ks2.get("attn", {}).get("tflops"),
                "whole_step_mfma_frac": fl2["total"] / (ms2 * 1e-3) / (PEAK_BF16_TFLOPS * 1e12)}
        if cached is not None:
            _, dt3, t3, fin3 = cached
            ms3 = 1e3 * dt3 / args.steps
            line["adapter_cached"] = {
                "variant": "text+layout with the ImageAdapter residuals computed once per prepare() through the fp32 path and kept across "
                           "the denoise steps (model.cache_adapter_residuals, the model class's default): the same function of the same "
                           "step-invariant input, 40-step parity 1.5e-3 instead of 1.3e-2 (profiles/r4c_gpu_parity.log); timed after "
                           "the headline with the same --steps / --warmup",
                "value": n_samples * args.steps / dt3, "unit": "denoise-steps/s", "ms_per_step": ms3, "flop_per_step": fl["total"],
                "finite": fin3, "gemm_tflops": t3.summary().get("gemm", {}).get("tflops"),
                "whole_step_mfma_frac": fl["total"] / (ms3 * 1e-3) / (PEAK_BF16_TFLOPS * 1e12)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(min(os.cpu_count() or 1, int(os.environ.get("DWM_CPU_THREADS", "64"))), args.layout, step_flop)
        print(json.dumps(line))
    D.shutdown()


if __name__ == "__main__":
    main()
