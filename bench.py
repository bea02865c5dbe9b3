"""Headline benchmark: 512x512 images/sec, SD-v1-4 architecture, 50-step PLMS, CFG 7.5 (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one image per GPU: `PLMSSamplerHIP.sample` (51 UNet calls through libsdmi's gfx950 kernels, fused
CFG + PLMS update) followed by the first-stage decode, also through libsdmi (`AutoencoderKLHIP`, SURVEY.md 8 f-1;
`--vae torch` runs the decoder on stock PyTorch-ROCm instead, for A/B).  Weights are
seeded random tensors in the exact SD-v1 architecture and the conditioning is synthetic (there is no checkpoint /
CLIP vocabulary in the environment); throughput is value independent.  N > 1: one process per GPU, one prompt per
GPU (weak scaling), no data-path collective; the finished latents are all-gathered once per step (RCCL).

Rank 0 prints ONE JSON line.  At N = 1 it also carries
  "roofline":     achieved TFLOP/s of the dominant kernel class, timed live with HIP events on the launch stream
                  (sdmi_profile_begin/end) over one UNet call of the same workload, vs the dense fp16 MFMA peak;
  "cpu_baseline": the oracle (CPU restatement of the reference, fp32, all host cores) timed on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
# the host driver only supports dmabuf IPC: RCCL needs this before HIP initialises (already exported on the GPU boxes)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

# profiler class-name prefixes of the GEMM family (one MFMA core + shared epilogue); the row-strip chain kernels (rowchain.hip: st_head = GroupNorm-apply
# + proj_in + q|k|v, st_mid = out-projection + to_q, st_tail / ff_tail = out-projection + GEGLU + FF-out + proj_out, gnconv3 = GroupNorm + conv3x3)
# run several of the reference's GEMMs per launch: their algorithmic flops are the sum of those GEMMs'
GEMM_CLASSES = ('igemm', 'conv3halo', 'gemm_split16', 'ff_tail', 'st_tail', 'st_head', 'st_mid', 'gnconv3')
MFMA_PEAK_TFLOPS = 2500.0     # dense fp16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
UNET_GFLOP = {64: 1606.5, 96: 4296.2}     # one UNet call, CFG batch 2, latent 64x64 / 96x96

# BASELINE.json configs: [1] is the headline (and the default); [3] and [4] are measured with --workload
WORKLOADS = {
    'txt2img512': dict(latent=64, sampler='plms', steps=50, scale=7.5, calls=51, vae_parts=1,
                       metric='512x512 images/sec, SD-v1-4 50-step PLMS CFG=7.5',
                       desc='latent 4x64x64 (512x512), 50 PLMS steps = 51 UNet calls at CFG batch 2, scale 7.5'),
    'txt2img768': dict(latent=96, sampler='ddim', steps=50, scale=7.5, calls=50, vae_parts=1,
                       metric='768x768 images/sec, SD-v1-4 50-step DDIM CFG=7.5',
                       desc='latent 4x96x96 (768x768, 9216-token self-attention), 50 DDIM steps = 50 UNet calls at CFG '
                            'batch 2, scale 7.5 (scripts/txt2img.py --H 768 --W 768, BASELINE.json configs[3])'),
    'img2img512': dict(latent=64, sampler='img2img', steps=50, scale=5.0, calls=37, vae_parts=3, strength=0.75,
                       metric='512x512 img2img images/sec, SD-v1-4 strength 0.75 of 50 DDIM steps CFG=5.0',
                       desc='first-stage encode of a 512x512 image, stochastic_encode at t_enc = 37, 37 DDIM steps at CFG '
                            'batch 2, scale 5.0 (scripts/img2img.py defaults, BASELINE.json configs[4]; the reference img2img '
                            'rejects --plms, img2img.py:205-207)'),
}


def build_gpu_model(device, seed=0, vae_kind='hip', vae_parts=1):
    from stable_diffusion_amd import AutoencoderKLHIP, LatentDiffusionHIP, UNetModelHIP
    from stable_diffusion_amd.synthetic import SD_V1_UNET_KWARGS, SD_V1_VAE_DDCONFIG, randomize_, randomize_vae_
    unet = UNetModelHIP(**SD_V1_UNET_KWARGS)
    ld = LatentDiffusionHIP(unet).to(device).eval()
    randomize_(unet, seed)
    if vae_kind == 'hip':       # the first stage on the same gfx950 kernels (decoder part only)
        vae = AutoencoderKLHIP(SD_V1_VAE_DDCONFIG, None, 4, parts=vae_parts).to(device).eval()
        randomize_vae_(vae, seed)
    else:                       # A/B: the decoder on stock PyTorch-ROCm (fp16 autocast), what north_star started from
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        from vae_torch import AutoencoderKLDecoder      # (tools/vae_torch.py: not part of the package)
        torch.manual_seed(seed)
        vae = AutoencoderKLDecoder().to(device).eval()
    return ld, unet, vae


def decode(vae, lat):
    """decode_first_stage + the script's clamp to [0, 1] (ddpm.py:705-763, scripts/txt2img.py:313-314)"""
    if hasattr(vae, '_handle'):
        img = vae.decode_first_stage(lat)
    else:
        with torch.autocast('cuda', dtype=torch.float16):
            img = vae.decode_first_stage(lat)
    return torch.clamp((img.float() + 1.0) / 2.0, min=0.0, max=1.0)


def one_image(sampler, vae, c, uc, x_T, steps_plms=50, scale=7.5):
    lat, _ = sampler.sample(S=steps_plms, batch_size=1, shape=list(x_T.shape[1:]), conditioning=c, verbose=False,
                            x_T=x_T, unconditional_guidance_scale=scale, unconditional_conditioning=uc, eta=0.0)
    return lat, decode(vae, lat)


def one_image_ddim(sampler, vae, c, uc, x_T, steps=50, scale=7.5):
    lat, _ = sampler.sample(S=steps, batch_size=1, shape=list(x_T.shape[1:]), conditioning=c, verbose=False,
                            x_T=x_T, unconditional_guidance_scale=scale, unconditional_conditioning=uc, eta=0.0)
    return lat, decode(vae, lat)


def one_image_img2img(sampler, vae, c, uc, init_image, steps=50, strength=0.75, scale=5.0):
    """scripts/img2img.py:229-262: encode_first_stage -> stochastic_encode(t_enc) -> decode(t_enc DDIM steps) -> decode_first_stage"""
    z0 = vae.encode_first_stage(init_image)                       # get_first_stage_encoding(encode_first_stage(x)): 0.18215 * sample
    sampler.make_schedule(ddim_num_steps=steps, ddim_eta=0.0, verbose=False)
    t_enc = int(strength * steps)
    z_enc = sampler.stochastic_encode(z0, torch.tensor([t_enc] * z0.shape[0], device=z0.device))
    lat = sampler.decode(z_enc, c, t_enc, unconditional_guidance_scale=scale, unconditional_conditioning=uc)
    return lat, decode(vae, lat)


def vae_latency_ms(vae, device, iters=5, H=64):
    g = torch.Generator(device='cpu').manual_seed(4)
    lat = (torch.randn(1, 4, H, H, generator=g) * 0.9).to(device)
    decode(vae, lat)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        decode(vae, lat)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def profile_unet(unet, device, H=64, W=64):
    """One UNet call (CFG batch 2) with per-launch HIP-event timing; returns the per-kernel-class table."""
    from stable_diffusion_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device='cpu').manual_seed(3)
    x = torch.randn(2, 4, H, W, generator=g).to(device)
    ctx = (0.1 * torch.randn(2, 77, 768, generator=g)).to(device)
    t = torch.tensor([481, 481], device=device)
    unet(x, t, context=ctx)
    torch.cuda.synchronize()
    _lib.check(lib.sdmi_profile_begin())
    unet(x, t, context=ctx)
    buf = bytes(1 << 16)
    import ctypes
    cbuf = ctypes.create_string_buffer(1 << 16)
    _lib.check(lib.sdmi_profile_end(cbuf, 1 << 16))
    return json.loads(cbuf.value.decode())


def unet_latency_ms(unet, device, H=64, W=64, iters=10):
    g = torch.Generator(device='cpu').manual_seed(4)
    x = torch.randn(2, 4, H, W, generator=g).to(device)
    ctx = (0.1 * torch.randn(2, 77, 768, generator=g)).to(device)
    t = torch.tensor([481, 481], device=device)
    unet.pin_context(ctx)
    for _ in range(2):
        unet(x, t, context=ctx)
    # host time to ENQUEUE one call, from an idle stream (round 6: the figure of rounds 1-5 was taken inside the 10-call loop below,
    # where the host blocks in hipLaunchKernel on a full queue -- it measured the device, 3.6 ms, not the executor: tools/host_enqueue.py)
    enq = []
    for _ in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        unet(x, t, context=ctx)
        enq.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        unet(x, t, context=ctx)
    t_host = time.perf_counter() - t0          # (with queue back-pressure: ~ the device time once the queue is full)
    torch.cuda.synchronize()
    unet.unpin_context()
    unet_latency_ms.host_enqueue_ms = sorted(enq)[len(enq) // 2] * 1e3
    unet_latency_ms.host_loop_ms = t_host / iters * 1e3
    return (time.perf_counter() - t0) / iters * 1e3


def box_probe(device):
    """What kind of box is this?  The pool's MI355X boxes differ by +-12 % on one binary, almost entirely on the latency-bound
    launches (rounds 2-4), so a bench line alone cannot tell a slow box from a regression.  Three numbers,
    ~50 ms of GPU time, all through the product library:
      empty_launch_us   back-to-back launches of a one-block kernel in one stream (the dependent-launch boundary);
      attn_d160_ctx_us  one fixed latency-bound launch of the UNet: cross-attention of the 16 x 16 level (16 heads x 256 queries, 77 keys, d = 160);
      conv_tflops       one fixed MFMA-bound launch: the 3 x 3 conv 320 -> 320 of the 64 x 64 level at CFG batch 8 (M = 32768);
      sclk_mhz          the shader clock the driver reports while that conv loop is queued (sysfs pp_dpm_sclk; null if unreadable)."""
    import ctypes
    import glob
    from stable_diffusion_amd import _lib
    lib = _lib.load()
    s_ = _lib.stream_ptr()

    def timed(fn, n, warm=10):
        for _ in range(warm):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    out = {}
    buf = torch.zeros(64, device=device)
    out['empty_launch_us'] = round(timed(lambda: _lib.check(lib.sdmi_k_prefetch_lines(buf.data_ptr(), 128, s_)), 500, 50), 3)
    g = torch.Generator(device='cpu').manual_seed(9)
    BH, heads, nq, nkv, d = 16, 8, 256, 77, 160
    q = (torch.randn(BH, nq, d, generator=g) * 0.3).half().to(device)
    k = (torch.randn(BH, nkv, d, generator=g) * 0.3).half().to(device)
    vt = torch.zeros(BH, d, 80, dtype=torch.float16, device=device)
    vt[:, :, :nkv] = (torch.randn(BH, d, nkv, generator=g) * 0.3).half().to(device)
    ao = torch.empty(BH // heads, nq, heads * d, dtype=torch.float16, device=device)
    out['attn_d160_ctx_us'] = round(timed(lambda: _lib.check(lib.sdmi_k_attention(q.data_ptr(), k.data_ptr(), vt.data_ptr(), ao.data_ptr(), BH, heads,
                                                                                   nq, nkv, 80, d, d ** -0.5, s_)), 300, 30), 3)
    B, H, C_ = 8, 64, 320
    a = (torch.randn(B * H * H, C_, generator=g) * 0.5).half().to(device)
    w = (torch.randn(C_, 9 * C_, generator=g) * 0.02).half().to(device)
    o = torch.empty(B * H * H, C_, device=device)
    dsc = _lib.IGemmDesc()
    dsc.a0 = a.data_ptr(); dsc.c0 = C_; dsc.lda0 = C_
    dsc.B, dsc.Hin, dsc.Win, dsc.Hout, dsc.Wout, dsc.ksize, dsc.stride = B, H, H, H, H, 3, 1
    dsc.w = w.data_ptr(); dsc.N = C_; dsc.out_f32 = o.data_ptr(); dsc.ldo = C_; dsc.splitk = 1; dsc.tile = -1; dsc.dma = -1
    conv = lambda: _lib.check(lib.sdmi_k_igemm(ctypes.byref(dsc), s_))
    us = timed(conv, 60, 10)
    out['conv_tflops'] = round(2.0 * B * H * H * C_ * 9 * C_ / us * 1e-6, 1)
    sclk = None
    try:
        for _ in range(200):          # ~25 ms of queued MFMA work; read the clock while it runs
            conv()
        for f in sorted(glob.glob('/sys/class/drm/card*/device/pp_dpm_sclk')):
            for line in open(f).read().splitlines():
                if line.strip().endswith('*'):
                    sclk = max(sclk or 0, int(line.split(':')[1].strip().lower().replace('mhz', '').replace('*', '').strip()))
    except Exception:
        sclk = None
    torch.cuda.synchronize()
    out['sclk_mhz'] = sclk
    # Round 5: the three probes above run with hot caches and do NOT separate the pool's box classes any more (three boxes at 5.2 / 6.5 / 6.6 ms per
    # UNet call showed the same 9.4 us / 510 TFLOP/s / 3 us).  Two HBM-side probes (they turned out not to separate the classes either) and a
    # kernel-switching probe (which does: profiles/bench_r05_boxes.txt):
    #   hbm_copy_gbs        a 1 GiB device-to-device copy (read + write), GB/s;
    #   cold_conv_us        the weight-streaming class: the 3 x 3 conv 1280 -> 1280 of the 8 x 8 level (M = 128, 29.5 MB of weights, split-K + reduce)
    #                       over 12 distinct weight buffers round-robin (354 MB: nothing survives in the 256 MB Infinity Cache);
    #   cold_conv_hot_us    the same launch on one buffer (weights cache-resident), for the ratio
    try:
        big_a = torch.empty(1 << 28, dtype=torch.float32, device=device); big_b = torch.empty_like(big_a)
        big_a.fill_(1.0)
        us = timed(lambda: big_b.copy_(big_a), 5, 2)
        out['hbm_copy_gbs'] = round(2.0 * big_a.numel() * 4 / us * 1e-3, 0)
        del big_a, big_b
        Bc, Hc, Cc = 2, 8, 1280
        a2 = (torch.randn(Bc * Hc * Hc, Cc, generator=g) * 0.5).half().to(device)
        w0 = (torch.randn(Cc, 9 * Cc, generator=g) * 0.01).half().to(device)
        ws = [w0] + [w0.clone() for _ in range(11)]
        o2 = torch.empty(Bc * Hc * Hc, Cc, device=device)
        wsp = torch.empty(16 * (Bc * Hc * Hc + 255) * (Cc + 255), dtype=torch.float32, device=device)
        d2 = _lib.IGemmDesc()
        d2.a0 = a2.data_ptr(); d2.c0 = Cc; d2.lda0 = Cc
        d2.B, d2.Hin, d2.Win, d2.Hout, d2.Wout, d2.ksize, d2.stride = Bc, Hc, Hc, Hc, Hc, 3, 1
        d2.N = Cc; d2.out_f32 = o2.data_ptr(); d2.ldo = Cc; d2.splitk = 0; d2.tile = -1; d2.dma = -1
        d2.splitk_ws = wsp.data_ptr(); d2.splitk_ws_floats = wsp.numel()
        state = {'i': 0}

        def cold():
            d2.w = ws[state['i'] % len(ws)].data_ptr(); state['i'] += 1
            _lib.check(lib.sdmi_k_igemm(ctypes.byref(d2), s_))

        def hot():
            d2.w = ws[0].data_ptr()
            _lib.check(lib.sdmi_k_igemm(ctypes.byref(d2), s_))
        out['cold_conv_us'] = round(timed(cold, 120, 24), 2)
        out['cold_conv_hot_us'] = round(timed(hot, 120, 24), 2)
        # ... and a CODE-side probe: the same small GEMM (M 512, N 1280, K 1280, operands cache-resident) through 18 different tile instantiations
        # round-robin (every launch starts with another kernel's code: instruction-cache-cold starts, as inside a UNet call, where ~40 kernels
        # alternate) against one instantiation repeated
        a3 = (torch.randn(512, 1280, generator=g) * 0.5).half().to(device)
        w3 = (torch.randn(1280, 1280, generator=g) * 0.02).half().to(device)
        o3 = torch.empty(512, 1280, device=device)
        d3 = _lib.IGemmDesc()
        d3.a0 = a3.data_ptr(); d3.c0 = 1280; d3.lda0 = 1280
        d3.B, d3.Hin, d3.Win, d3.Hout, d3.Wout, d3.ksize, d3.stride = 1, 512, 1, 512, 1, 1, 1
        d3.w = w3.data_ptr(); d3.N = 1280; d3.out_f32 = o3.data_ptr(); d3.ldo = 1280; d3.splitk = 1; d3.dma = -1
        tiles = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 18, 19, 20, 21]
        st3 = {'i': 0}

        def mixed():
            d3.tile = tiles[st3['i'] % len(tiles)]; st3['i'] += 1
            _lib.check(lib.sdmi_k_igemm(ctypes.byref(d3), s_))

        def same():
            d3.tile = 5
            _lib.check(lib.sdmi_k_igemm(ctypes.byref(d3), s_))
        out['mixed_kernels_us'] = round(timed(mixed, 180, 36), 2)
        out['same_kernel_us'] = round(timed(same, 180, 36), 2)
        # the same 18 instantiations, each repeated 10 x in a row (hot code): the average that mixed_kernels_us is to be compared with
        tot = 0.0
        for tl in tiles:
            d3.tile = tl
            tot += timed(lambda: _lib.check(lib.sdmi_k_igemm(ctypes.byref(d3), s_)), 10, 4) * 10
        out['mixed_set_hot_us'] = round(tot / (10 * len(tiles)), 2)
    except Exception as e:      # noqa: BLE001  (a probe must never fail the bench)
        out['hbm_probe_error'] = f'{type(e).__name__}: {e}'[:200]
    out['note'] = ('round 5, 14 boxes (profiles/bench_r05_boxes.txt): the hot probes (empty_launch / attn_d160_ctx / conv_tflops) and the HBM-side ones '
                   '(hbm_copy_gbs / cold_conv_us) read the same on the fast (5.2 - 5.4 ms per UNet call) and slow (6.4 - 6.6 ms) boxes of the pool; '
                   'mixed_kernels_us separates them (fast ~14.5 us, slow ~16.1 us; mixed_set_hot_us = the same kernels without switching), and so does '
                   'unet_host_enqueue_ms_per_call (fast ~3.5 ms, slow ~4.4 ms)')
    return out


def offline_traffic(kernel_class):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC pass (profiles/traffic_rNN.json, the newest);
    PMC counters cannot be collected inside this process, so `roofline.traffic` is read from that committed pass."""
    try:
        path = next(p_ for p_ in (os.path.join(ROOT, 'profiles', f'traffic_r0{r}.json') for r in (6, 5, 4, 3, 2)) if os.path.exists(p_))
        d = json.load(open(path))
        k = d['kernels'].get(kernel_class)
        if k:
            return {'gbytes_per_launch': round((k['fetch_mb_x2'] + k['write_mb']) / 1e3, 4), 'source': d['source'],
                    'note': 'FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md'}
    except Exception:
        pass
    return None


def traffic_pass(out_json=None, keep_dir=None, also=None):
    """`bench.py --traffic-pass`: HBM bytes per launch of every kernel class of one UNet call, measured: two rocprofv3
    PMC passes (FETCH_SIZE, WRITE_SIZE -- separate runs, no trace domains beside the kernel trace, as
    MI355X_MICROARCH.md prescribes) over tools/prof_shapes.py, FETCH_SIZE doubled (gfx950 counts 64-byte requests in
    its 32-byte unit), summed per kernel family and written to profiles/traffic_rNN.json, which the default run reads
    for `roofline.traffic`.  Needs rocprofv3 and a GPU; takes ~1 minute."""
    import glob
    import sqlite3
    import subprocess
    import tempfile
    out_json = out_json or os.path.join(ROOT, 'profiles', 'traffic_r06.json')
    work = keep_dir or tempfile.mkdtemp(prefix='sdmi_traffic_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp')
    per = {}
    for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = os.path.join(work, ctr)
        cmd = ['rocprofv3', '--pmc', ctr, '-d', d, '-o', 'pmc', '--', sys.executable, os.path.join(ROOT, 'tools', 'prof_shapes.py')]
        r = subprocess.run(cmd, env=env, cwd='/tmp', capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            raise SystemExit(f'rocprofv3 --pmc {ctr} failed:\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}')
        for f in glob.glob(os.path.join(d, '**', '*_results.db'), recursive=True):
            con = sqlite3.connect(f)
            q = ("select kernel_name, sum(value), count(distinct dispatch_id) from counters_collection "
                 f"where counter_name = '{ctr}' group by kernel_name")
            for name, total, launches in con.execute(q):
                e = per.setdefault(name, {'launches': 0, 'FETCH_SIZE': 0.0, 'WRITE_SIZE': 0.0})
                e[ctr] += float(total)            # rocprofv3 reports these two derived counters in KiB
                e['launches'] = max(e['launches'], int(launches))
            con.close()

    def family(name):
        for key, fam in (('igemm_kernel', 'igemm_family'), ('igemm5_kernel', 'igemm_family'), ('conv3halo_kernel', 'igemm_family'),
                         ('conv3halo_gn_kernel', 'igemm_family'), ('gemm_split16_kernel', 'igemm_family'), ('ff_tail_kernel', 'igemm_family'), ('st_head_kernel', 'igemm_family'), ('gn_conv3_kernel', 'igemm_family'),
                         ('attn', 'attention'),
                         ('splitk_reduce', 'splitk_reduce'), ('gn_apply', 'groupnorm'), ('gn_stats', 'groupnorm'),
                         ('layernorm', 'layernorm')):
            if key in name:
                return fam
        return None
    kernels = {}
    for name, e in per.items():
        fam = family(name)
        if fam is None or not e['launches']:
            continue
        k = kernels.setdefault(fam, {'launches': 0, 'fetch_kb_x2': 0.0, 'write_kb': 0.0})
        k['launches'] += e['launches']
        k['fetch_kb_x2'] += 2.0 * e['FETCH_SIZE']
        k['write_kb'] += e['WRITE_SIZE']
    for k in kernels.values():       # per launch, in MB (the keys offline_traffic() reads)
        k['fetch_mb_x2'] = k.pop('fetch_kb_x2') / 1024.0 / k['launches']
        k['write_mb'] = k.pop('write_kb') / 1024.0 / k['launches']
    doc = {'source': 'bench.py --traffic-pass: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over '
                     'tools/prof_shapes.py (model build + UNet calls, CFG batch 2, 64x64)', 'kernels': kernels}
    for path in [out_json] + ([also] if also else []):
        with open(path, 'w') as f:
            json.dump(doc, f, indent=1, sort_keys=True)
    print(json.dumps({'traffic_pass': out_json, 'kernels': kernels}))


def usable_cores():
    """Host cores this process may actually use: min(affinity mask, cgroup v2/v1 CPU quota) -- the GPU boxes report
    256 logical CPUs but run the job under a 16-CPU quota, where 256 torch threads are ~300x slower than 16."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def reference_unet():
    """The real reference UNetModel (ldm/modules/diffusionmodules/openaimodel.py): from its sources when they are visible (the
    build container has them under /root/reference or $SD_REFERENCE), else from the bytecode bundle oracle/build_ref_bundle.py
    compiled from them (oracle/_ref/refbundle: git-ignored, travels to the GPU box with the snapshot like the built .so --
    the same bundle tests/test_reference_script_gpu.py runs there).  Returns (callable (x, t, context) -> eps with the oracle's
    synthetic SD-v1 weights loaded strict=True, where it came from) or (None, None)."""
    ref = os.environ.get('SD_REFERENCE', '/root/reference')
    bundle = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'oracle', '_ref', 'refbundle')
    if os.path.isdir(os.path.join(ref, 'ldm')):
        root, origin = ref, 'sources'
    elif os.path.exists(os.path.join(bundle, 'ldm', 'modules', 'diffusionmodules', 'openaimodel.pyc')):
        root, origin = bundle, 'bytecode bundle oracle/_ref/refbundle'
    else:
        return None, None
    try:
        from oracle.make_golden import _import_reference
        from oracle.plan import SD_V1
        from oracle.weights import make_state_dict
        UNetModel = _import_reference(root)[0]
        m = UNetModel(**SD_V1.ref_kwargs()).eval()
        m.load_state_dict(make_state_dict(SD_V1, 0), strict=True)
        return (lambda x, t, c: m(x, t, context=c)), origin
    except Exception as e:        # a missing dependency of the reference: fall back to the restatement, and say so
        print(f'[bench] reference UNet not importable ({type(e).__name__}: {e}); CPU baseline uses the oracle port', file=sys.stderr)
        return None, None


def cpu_baseline(n_unet_calls=2):
    """The CPU comparator on the host cores: the reference's own UNetModel when /root/reference is visible (kind
    "reference"), else the oracle (fp32 CPU restatements of the reference UNet; kind "port"); the first-stage decode is
    the oracle's in both cases.  Bounded sample: `n_unet_calls` UNet calls at the full workload shape (CFG batch 2, latent
    64x64) and one VAE decode, extrapolated to 51 calls + 1 decode per image."""
    from oracle import unet_ref
    from oracle.plan import SD_V1
    from oracle.weights import make_inputs, make_state_dict
    from oracle import vae_ref
    cores = usable_cores()
    torch.set_num_threads(cores)
    x, t, ctx = make_inputs(SD_V1, 2, 64, 64, seed=1)
    ref_fn, ref_origin = reference_unet()
    kind = 'reference' if ref_fn is not None else 'port'
    if ref_fn is None:
        sd = make_state_dict(SD_V1, 0)
        ref_fn = lambda x_, t_, c_: unet_ref.unet_forward(sd, SD_V1, x_, t_, c_)
    times = []
    with torch.no_grad():
        for _ in range(n_unet_calls):
            t0 = time.perf_counter()
            ref_fn(x, t, ctx)
            times.append(time.perf_counter() - t0)
            if times[-1] > 20.0:          # keep the sample bounded on slow hosts
                break
    t_unet = min(times)
    vsd = vae_ref.make_vae_state_dict(vae_ref.SD_VAE, 0, encoder=False)
    z = vae_ref.make_vae_inputs(vae_ref.SD_VAE, 1, 64, 64, seed=1) * 0.18215
    t0 = time.perf_counter()
    vae_ref.decode_first_stage(vsd, vae_ref.SD_VAE, z)
    t_vae = time.perf_counter() - t0
    s_per_image = 51 * t_unet + t_vae
    who = f'reference UNetModel ({ref_origin})' if kind == 'reference' else 'oracle UNet'
    return {'value': 1.0 / s_per_image, 'unit': 'images/s', 'cores': cores, 'kind': kind,
            'sample': f'{len(times)} {who} call(s) (fp32, {cores} threads, B=2, 64x64 latent: {t_unet:.2f} s each) + 1 VAE decode '
                      f'({t_vae:.2f} s), extrapolated to 51 calls + 1 decode = {s_per_image:.1f} s/image'}


def timed_steps(step_fn, steps, warmup, world, rank, device):
    """The driver's timing contract: `warmup` untimed steps, then exactly `steps` steps bracketed by a barrier + device
    synchronize on both sides; returns (max-over-ranks seconds, last local latent, last image, last gathered latents).
    `step_fn(step) -> (latent, image)`; every rank all_gathers its finished latent once per step (the path's only
    collective).  Device-agnostic so tests/test_dist_gloo.py can run it with world_size 2 on the CPU."""
    import contextlib
    import io
    from stable_diffusion_amd import dist as sd_dist
    quiet = contextlib.redirect_stdout(io.StringIO())

    def sync():
        if device.type == 'cuda':
            torch.cuda.synchronize()
        if torch.distributed.is_available() and torch.distributed.is_initialized():     # (also a launched world of size 1)
            torch.distributed.barrier()
        if device.type == 'cuda':
            torch.cuda.synchronize()
    with quiet:
        for w in range(warmup):
            step_fn(-1 - w)
    sync()
    t0 = time.perf_counter()
    lat = img = allz = None
    with quiet:
        for s in range(steps):
            lat, img = step_fn(s)
            allz = sd_dist.gather_latents(lat, world, rank, world)     # 64 KiB per image
    sync()
    elapsed = sd_dist.max_over_ranks(time.perf_counter() - t0, device)
    return elapsed, lat, img, allz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--vae', choices=['hip', 'torch'], default='hip', help='first-stage decode: libsdmi (default) or PyTorch-ROCm')
    ap.add_argument('--workload', choices=sorted(WORKLOADS), default='txt2img512',
                    help='txt2img512 = BASELINE.json configs[1] (the headline metric, default); txt2img768 = configs[3]; '
                         'img2img512 = configs[4]')
    ap.add_argument('--traffic-pass', action='store_true',
                    help='measure HBM bytes per launch with rocprofv3 PMC passes and write profiles/traffic_r06.json (then exit)')
    ap.add_argument('--traffic-out', default=None, help='second copy of the --traffic-pass JSON (e.g. under gpurun_out/)')
    ap.add_argument('--cpu-baseline-only', action='store_true', help='time only the CPU comparator (no GPU needed) and exit')
    args = ap.parse_args()
    if args.traffic_pass:
        return traffic_pass(also=args.traffic_out)
    if args.cpu_baseline_only:
        print(json.dumps({'cpu_baseline': cpu_baseline(1)}))
        return
    wl = WORKLOADS[args.workload]
    LAT = wl['latent']

    from stable_diffusion_amd import DDIMSamplerHIP, PLMSSamplerHIP
    from stable_diffusion_amd import dist as sd_dist
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the product path has no CPU fallback')
    rank, world, local_rank = sd_dist.init_from_env()
    torch.set_num_threads(max(1, usable_cores() // max(1, world)))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    device = torch.device('cuda', local_rank if world > 1 else 0)
    torch.cuda.set_device(device)

    if not args.no_roofline:
        os.environ.setdefault('SDMI_PROF_SHAPES', '1')     # per-shape class names in the profiling call (read once by libsdmi)
    ld, unet, vae = build_gpu_model(device, vae_kind=args.vae, vae_parts=wl['vae_parts'])
    sampler = PLMSSamplerHIP(ld) if wl['sampler'] == 'plms' else DDIMSamplerHIP(ld)
    # synthetic conditioning / start codes; the seed depends on the GLOBAL prompt index only
    def inputs(step):
        gidx = step * world + rank
        g = torch.Generator(device='cpu').manual_seed(1000 + gidx)
        c = (0.1 * torch.randn(1, 77, 768, generator=g)).to(device)
        if wl['sampler'] == 'img2img':       # a synthetic init image in [-1, 1] (img2img.py:49-60 load_img)
            x_T = (torch.rand(1, 3, LAT * 8, LAT * 8, generator=g) * 2.0 - 1.0).to(device)
        else:
            x_T = torch.randn(1, 4, LAT, LAT, generator=g).to(device)
        return c, x_T
    guc = torch.Generator(device='cpu').manual_seed(2)
    uc = (0.1 * torch.randn(1, 77, 768, generator=guc)).to(device)

    def inputs_for(step):
        c, x_T = inputs(step)
        return c, uc, x_T

    if wl['sampler'] == 'plms':
        step_fn = lambda step: one_image(sampler, vae, *inputs_for(step), steps_plms=wl['steps'], scale=wl['scale'])
    elif wl['sampler'] == 'ddim':
        step_fn = lambda step: one_image_ddim(sampler, vae, *inputs_for(step), steps=wl['steps'], scale=wl['scale'])
    else:
        step_fn = lambda step: one_image_img2img(sampler, vae, *inputs_for(step), steps=wl['steps'],
                                                 strength=wl['strength'], scale=wl['scale'])
    elapsed, lat, img, allz = timed_steps(step_fn, args.steps, args.warmup, world, rank, device)
    assert torch.isfinite(img).all() and torch.isfinite(allz).all()
    ranks = sd_dist.ranks_seen(device)          # (a collective: every rank calls it)
    assert ranks == world, f'{ranks} ranks answered, WORLD_SIZE = {world}'

    out = None
    if rank == 0:
        n_images = args.steps * world
        out = {
            'metric': wl['metric'],
            'value': n_images / elapsed, 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f16', 'data': 'synthetic',
            'config': {'workload': f'{args.workload}: SD-v1-4 UNet (859.5M params, random init), ' + wl['desc'] +
                                   ', 1 prompt per GPU per step, + VAE decode ' +
                                   ('(libsdmi AutoencoderKLHIP)' if args.vae == 'hip' else '(PyTorch-ROCm fp16 autocast)'),
                       'global_batch': world, 'parallelism': f'dp{world} (one prompt per GPU, latents all_gather)'},
        }
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            # (a process group exists whenever a launcher exported RANK / WORLD_SIZE, also at world size 1)
            out['collective'] = {'backend': torch.distributed.get_backend(), 'world_size': torch.distributed.get_world_size(),
                                 'ops_per_step': 'one all_gather of the finished latents (64 KiB per image)',
                                 'timing': 'all_reduce(MAX) of the timed region over the ranks'}
        out['ranks_seen'] = ranks
        if world == 1:
            out['box_probe'] = box_probe(device)
            ms = unet_latency_ms(unet, device, H=LAT, W=LAT)
            out['unet_ms_per_call'] = ms
            out['unet_host_enqueue_ms_per_call'] = round(getattr(unet_latency_ms, 'host_enqueue_ms', float('nan')), 3)   # host time to enqueue one call from an idle stream (launch tapes, csrc/tape.h)
            out['unet_host_loop_ms_per_call'] = round(getattr(unet_latency_ms, 'host_loop_ms', float('nan')), 3)       # ... inside a back-to-back loop: queue back-pressure included (the round 1-5 figure)
            out['unet_calls_per_image'] = wl['calls']
            out['unet_tflops'] = UNET_GFLOP[LAT] / ms
            out['vae_decode_ms'] = vae_latency_ms(vae, device, H=LAT)
            if not args.no_roofline:
                table = profile_unet(unet, device, H=LAT, W=LAT)
                table.sort(key=lambda r: -r['ms'])
                # The implicit-GEMM kernel is ONE template (csrc/igemm.hip) launched in several tile instantiations
                # chosen per shape by the tuning table: the dominant kernel is that family; its instantiations are listed.
                # Timing every launch with HIP events inflates it (round 4: per-class sum 7.32 ms vs 6.59 ms for the un-profiled call,
                # +13 % vs rocprofv3 kernel durations).  The table is therefore SCALED by (un-profiled call) / (sum of the event-timed
                # classes) -- the figure that agrees with `rocprofv3 --kernel-trace --stats` of the same call to ~2 % -- and the raw
                # event figures are kept beside it (`*_events`).
                # the event overhead is roughly CONSTANT per launch, so a uniform scale over-corrects the long GEMM launches
                # and under-corrects the short helpers.  Round 6: the primary correction subtracts one constant per launch,
                # c = (sum of event-timed launches - un-profiled call) / launches; the uniformly scaled and the raw figures stay beside it.
                ev_sum = sum(r['ms'] for r in table)
                n_launch = sum(r['launches'] for r in table)
                ev_scale = min(1.0, ms / ev_sum) if ev_sum > 0 else 1.0
                ev_const = max(0.0, (ev_sum - ms) / n_launch) if n_launch else 0.0
                for r in table:
                    r['ms_events'] = r['ms']
                    r['ms_scaled'] = r['ms'] * ev_scale
                    r['ms'] = max(r['ms'] - ev_const * r['launches'], 0.25 * r['ms'])
                # the st_mid chain launch runs the cross-attention too: its attention flops do not belong to the GEMM family
                for r in table:
                    if r['name'].startswith('st_mid_ctx'):
                        gemm_fl = r['launches'] * 2.0 * (2 * LAT * LAT) * (2.0 * 320 * 320)
                        r['flops_attn'] = max(0.0, r['flops'] - gemm_fl)
                        r['flops'] = min(r['flops'], gemm_fl)
                        if 'flops_exec' in r:
                            r['flops_exec'] = max(0.0, r['flops_exec'] - r['flops_attn'])
                fam = [r for r in table if r['name'].startswith(GEMM_CLASSES)]
                dom = {'name': 'igemm_kernel / conv3halo_kernel / gemm_split16_kernel / ff_tail_kernel / st_head_kernel (the GEMM family: one MFMA core + shared epilogue, all tile instantiations; the row-strip chain kernels run 2-4 of the reference GEMMs per launch)',
                       'launches': sum(r['launches'] for r in fam), 'ms': sum(r['ms'] for r in fam),
                       'flops': sum(r['flops'] for r in fam), 'flops_exec': sum(r.get('flops_exec', r['flops']) for r in fam),
                       'bytes': sum(r['bytes'] for r in fam)}
                top = fam[0]
                mfma = [r for r in table if r['flops'] > 0 and r['name'].startswith(GEMM_CLASSES + ('attn',))]      # (st_mid_ctx: GEMM flops only)
                # `flops` are ALGORITHMIC (2 x MACs of the reference's convs / linears); the 3-pass split-fp16
                # 1x1 convs execute 3x theirs, which only `achieved_executed` / `frac_executed` count
                ach = dom['flops'] / (dom['ms'] * 1e-3) / 1e12
                ach_exec = dom['flops_exec'] / (dom['ms'] * 1e-3) / 1e12
                cls = lambda r: {'name': r['name'], 'launches': r['launches'], 'ms': round(r['ms'], 4),
                                 'tflops': round(r['flops'] / (r['ms'] * 1e-3) / 1e12, 1) if r['flops'] else None,
                                 'gbs': round(r['bytes'] / (r['ms'] * 1e-3) / 1e9, 1)}
                out['roofline'] = {
                    'bound': 'mfma', 'kernel': dom['name'], 'launches_per_unet_call': dom['launches'],
                    'avg_launch_ms': dom['ms'] / dom['launches'],
                    'avg_launch_ms_events': sum(r['ms_events'] for r in fam) / dom['launches'],
                    'avg_launch_ms_scaled': sum(r['ms_scaled'] for r in fam) / dom['launches'],
                    'frac_events': dom['flops'] / (sum(r['ms_events'] for r in fam) * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                    'frac_scaled': dom['flops'] / (sum(r['ms_scaled'] for r in fam) * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                    'timing': f'per-launch HIP events on the launch stream minus a constant {ev_const * 1e3:.2f} us per launch = (sum of the event-timed '
                              f'launches ({ev_sum:.3f} ms) - un-profiled UNet call ({ms:.3f} ms)) / {n_launch} launches; frac_events = raw events, '
                              f'frac_scaled = round 5\'s uniform scale {ev_scale:.4f}; compare with rocprofv3 --kernel-trace --stats (profiles/kernel_stats_bench_r06.txt)',
                    'achieved': ach, 'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / MFMA_PEAK_TFLOPS,
                    'achieved_executed': ach_exec, 'frac_executed': ach_exec / MFMA_PEAK_TFLOPS,
                    'algorithmic_gflop_per_launch': dom['flops'] / dom['launches'] / 1e9,
                    'share_of_unet_call': dom['ms'] / sum(r['ms'] for r in table),
                    'top_instantiation': dict(cls(top), frac=round(top['flops'] / (top['ms'] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)),
                    # HBM bytes per launch from the committed rocprofv3 PMC passes over the same UNet call (separate
                    # FETCH_SIZE / WRITE_SIZE passes cannot run inside this process); null if absent
                    'traffic': (lambda t: t['gbytes_per_launch'] * 1e9 if t else None)(offline_traffic('igemm_family')),
                    'traffic_unit': 'bytes/launch', 'traffic_offline': offline_traffic('igemm_family'),
                    'algorithmic_gbytes_per_launch': dom['bytes'] / dom['launches'] / 1e9,
                    'per_class': [cls(r) for r in table],
                    'mfma_classes_tflops': sum(r['flops'] for r in mfma) / (sum(r['ms'] for r in mfma) * 1e-3) / 1e12
                    if mfma else None,
                }
                # per-class table: the GEMM shapes folded back into their tile instantiation (the per-shape names are
                # only needed for the weight-streaming entry below)
                import re as _re
                folded = {}
                for r in table:
                    key = _re.sub(r'_M\d+_N\d+_K\d+.*$', '', r['name'])
                    f_ = folded.setdefault(key, {'name': key, 'launches': 0, 'ms': 0.0, 'flops': 0.0, 'flops_exec': 0.0, 'bytes': 0.0})
                    for k_ in ('launches', 'ms', 'flops', 'flops_exec', 'bytes'):
                        f_[k_] += r.get(k_, r['flops'] if k_ == 'flops_exec' else 0)
                out['roofline']['per_class'] = [cls(r) for r in sorted(folded.values(), key=lambda r: -r['ms'])]
                # the weight-streaming GEMMs (M <= 128 rows, i.e. the 8x8 level: every weight byte is a first touch and
                # there are only 0.13 MFLOP per weight byte): bytes = weights + activations + outputs, against HBM peak
                ws = [r for r in fam if (lambda m: m and int(m.group(1)) <= 128)(_re.search(r'_M(\d+)_N', r['name']))]
                if ws:
                    wb, wms, wn = sum(r['bytes'] for r in ws), sum(r['ms'] for r in ws), sum(r['launches'] for r in ws)
                    gbs_w = wb / (wms * 1e-3) / 1e9
                    # ... and END TO END, with the split-K reduction launch behind every split GEMM (the reduce
                    # class is not named per shape: its average launch time x the number of split launches among these shapes)
                    red = [r for r in table if r['name'].startswith('splitk_reduce')]
                    red_avg = (sum(r['ms'] for r in red) / max(1, sum(r['launches'] for r in red))) if red else 0.0
                    n_split = sum(r['launches'] for r in ws if (lambda m: m and int(m.group(1)) > 1)(_re.search(r'_s(\d+)$', r['name'])))
                    wms_all = wms + n_split * red_avg
                    out['roofline_weight_stream'] = {
                        'bound': 'hbm', 'kernel': 'igemm_kernel / conv3halo_kernel on the M <= 128 shapes (the 8x8 level: split-K partial GEMMs)',
                        'launches_per_unet_call': wn, 'avg_launch_ms': wms / wn, 'achieved': gbs_w, 'peak': HBM_PEAK_GBS,
                        'unit': 'GB/s', 'frac': gbs_w / HBM_PEAK_GBS, 'traffic': None,
                        'algorithmic_gbytes_per_launch': wb / wn / 1e9,
                        'split_launches': n_split, 'reduce_avg_launch_ms': red_avg,
                        'avg_ms_with_reduce': wms_all / wn, 'achieved_with_reduce': wb / (wms_all * 1e-3) / 1e9,
                        'frac_with_reduce': wb / (wms_all * 1e-3) / 1e9 / HBM_PEAK_GBS}
                # second entry: the dominant HBM-bound kernel class (norms / reduce / casts), against the HBM peak
                hbm = [r for r in table if not r['name'].startswith(GEMM_CLASSES + ('attn',))]
                if hbm:
                    h = hbm[0]
                    gbs = h['bytes'] / (h['ms'] * 1e-3) / 1e9
                    # measured HBM bytes per launch of that class from the committed PMC pass (same source as roofline.traffic)
                    t_hbm = offline_traffic(h['name']) if h['name'] in ('groupnorm', 'splitk_reduce', 'layernorm') else None
                    out['roofline_hbm'] = {'bound': 'hbm', 'kernel': h['name'], 'launches_per_unet_call': h['launches'],
                                           'avg_launch_ms': h['ms'] / h['launches'], 'achieved': gbs, 'peak': HBM_PEAK_GBS,
                                           'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS,
                                           'traffic': t_hbm['gbytes_per_launch'] * 1e9 if t_hbm else None,
                                           'algorithmic_gbytes_per_launch': h['bytes'] / h['launches'] / 1e9}
            if not args.no_cpu_baseline and args.workload == 'txt2img512':     # the CPU comparator is quoted on the headline config
                out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out))
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
