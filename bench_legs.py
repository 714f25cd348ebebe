"""bench_legs.py -- the legs of bench.py's N = 1 line BEHIND the timed region of the contract leg: BASELINE configs[2] (base-band
I/Q, n = 1024), configs[3] (UC_STREAM from a captured hipGraph), hello_world1 (configs[4] through the C group at world size 1),
the multi-stream receivers (recorded, live, PDM), and the side measurements of the sibling kernels (`bench.py --variant ...`).
Split from bench.py in round 6; bench.py re-exports everything here."""
import json
import os
import sys
import time

import numpy as np

from bench_telemetry import *  # noqa: F401,F403  (constants, roofline(), timed_launches(), clock helpers ...)

def config2_iq(args, device, stream, torch, mag_mean):
    """BASELINE configs[2] on its stated workload: a continuous real pass-band stream at
    fs = 100 kHz, carrier 18 kHz, base-band chirps of +-1.5 kHz, n = 1024 samples per symbol, 26 samples of FIR
    history in front, generated on the device; UC_IQ with UC_FLAG_IQ_BASEBAND (the experiment's intended maths,
    simulation/IQ_modulation.ipynb cells 16-31: mix, 27-tap low-pass, two dechirp runs, up/down symbol) and, beside
    it, the firmware-window mode (experiments/iq_modulation/Src/main.c:283-285: one history, no symbol)."""
    import uchirp
    from uchirp import synth
    n = 1024
    nf = 2 * args.frames                                  # the same sample count as configs[1]
    x, bits = synth.device_iq_stream(nf, n, device, seed=4321, snr_db=args.snr)
    steps, warm = min(args.steps, 10), min(args.warmup, 3)
    cfg = dict(n=n, fs=100000.0, carrier=18000.0, f0=16500.0, f1=19500.0, time_frame=n / 100000.0, mag_mean=mag_mean)
    out = {"workload": "configs[2]: %d x %d-sample symbols of a continuous real pass-band stream, fs 100 kHz, carrier 18 kHz, "
                       "base-band chirp +-1.5 kHz, SNR %.0f dB, 26 samples of FIR history, generated on the device" % (nf, n, args.snr),
           "frames": nf, "frame_len": n, "steps": steps, "warmup": warm}
    # --- base-band mode: symbols out
    eng = uchirp.Engine(uchirp.IQ, device=device.index, flags=uchirp.FLAG_IQ_BASEBAND, **cfg)
    sym = torch.empty(nf, dtype=torch.uint8, device=device)

    def launch_bb():
        eng.process(x, n_frames=nf, want_stats=False, symbols_out=sym, stream=stream.cuda_stream)

    ramp = clock_ramp(launch_bb, torch, args.ramp_ms)
    wall, kern = timed_launches(launch_bb, stream, torch, steps, warm)
    ncu, clk = num_cus(torch, device), live_clock(eng, launch_bb)
    bb = {"mode": "UC_FLAG_IQ_BASEBAND: mix, 27-tap FIR, dechirp by conj(up) and conj(down), 2 x CFFT-1024, windows around DC, symbol",
          "value": nf / (wall * 1e-3), "unit": "frames/s", "ms_per_step": wall, "ramp_launches": ramp,
          "roofline": roofline("iq1024_bb_f32", "iq1024_kernel<f32,baseband>", nf, 4096 + 1, kern, ncu, clk),
          "bytes_note": "4096 B in + 1 B symbol out per frame (+ 104 B of FIR history once per launch)",
          "bit_error_rate_vs_transmitted": float((sym != bits).float().mean().item())}
    if not args.no_cpu_baseline:
        from oracle import uco
        o = uco.Oracle(uco.IQ, flags=uco.FLAG_IQ_BASEBAND, **cfg)
        head = x[: 26 + 4096 * n].cpu().numpy()
        rs, rst = o.process(head, halo=26, n_frames=4096)
        clear = clear_frames(rst)
        got = sym[:4096].cpu().numpy()
        # (all clear frames equal AND there are clear frames: the mean of an empty mask is NaN, which no comparison catches)
        bb["symbols_equal_oracle_head4096_clear"] = float((got[clear] == rs[clear]).mean()) if clear.any() else 0.0
        bb["oracle_head_clear_frames"] = int(clear.sum())
        bb["oracle_head_near_ties_excluded"] = int((~clear).sum())
        bb["oracle_head_bit_error_rate"] = float((rs != bits[:4096].cpu().numpy()).mean())
    out["baseband"] = bb
    eng.close()
    # --- firmware-window mode: one history record out, no symbol
    eng = uchirp.Engine(uchirp.IQ, device=device.index, n=n, mag_mean=mag_mean)
    st = torch.empty((nf, eng.spf, 8), dtype=torch.float32, device=device)

    def launch_fw():
        eng.process(x, n_frames=nf, want_symbols=False, stats_out=st, stream=stream.cuda_stream)

    clock_ramp(launch_fw, torch, args.ramp_ms / 3)
    wall, kern = timed_launches(launch_fw, stream, torch, steps, warm)
    clk = live_clock(eng, launch_fw)
    out["firmware_windows"] = {
        "mode": "the committed firmware's windows at bin (F1+F2) n / fs (iq_modulation/Src/main.c:215-219,283-285): one dechirp run, one history",
        "value": nf / (wall * 1e-3), "unit": "frames/s", "ms_per_step": wall,
        "roofline": roofline("iq1024_fw_f32", "iq1024_kernel<f32,firmware windows>", nf, 4096 + 32, kern, ncu, clk),
        "bytes_note": "4096 B in + 32 B history record out per frame"}
    out["baseband_over_firmware_windows"] = bb["value"] / out["firmware_windows"]["value"]
    eng.close()
    del x, sym, st
    return out


def config3_stream(args, frames, device, torch):
    """BASELINE configs[3]: the configs[1] batch read as ONE continuous stream through UC_STREAM (27-tap FIR low-pass,
    decimation by 8, overlap-save FFT x H x IFFT compression, include/uchirp.h), captured ONCE into a hipGraph and
    replayed; the eager launches beside it.  The captured launch deals its blocks dynamically like the eager one
    (a counter slot the graph owns)."""
    import uchirp
    eng = uchirp.Engine(uchirp.STREAM, device=device.index)
    x = frames.reshape(-1)
    halo, n_out, n_blocks, hop = eng.stream_geometry(x.numel())
    steps, warm = min(args.steps, 10), min(args.warmup, 3)
    comp = torch.empty(n_out, dtype=torch.float32, device=device)
    pk = torch.empty((n_blocks, 2), dtype=torch.int32, device=device)
    comp_g, pk_g = torch.empty_like(comp), torch.empty_like(pk)
    byts = (x.numel() * 4 + n_out * 4 + n_blocks * 8) / x.numel()
    out = {"workload": "configs[3]: %d samples (2^%.2f, the configs[1] batch as one continuous stream), decimation 8, FFT 2048, "
                       "hop %d, |y| for every decimated sample + one peak record per block" % (x.numel(), np.log2(x.numel()), hop),
           "samples": x.numel(), "blocks": n_blocks, "decim": int(eng.cfg.decim), "steps": steps, "warmup": warm,
           "bytes_per_sample": byts}
    s1 = torch.cuda.current_stream(device)

    def launch():
        eng.process_stream(x, compressed_out=comp, peaks_out=pk, stream=s1.cuda_stream)

    clock_ramp(launch, torch, args.ramp_ms)
    wall, kern = timed_launches(launch, s1, torch, steps, warm)
    ncu, clk = num_cus(torch, device), live_clock(eng, launch)
    out["eager"] = {"value": x.numel() / (wall * 1e-3), "unit": "samples/s", "ms_per_step": wall,
                    "roofline": roofline("stream_d8_f32", "stream_kernel<f32,8>", x.numel(), byts, kern, ncu, clk)}
    s2 = torch.cuda.Stream(device)
    g = torch.cuda.CUDAGraph()
    s2.wait_stream(s1)
    with torch.cuda.stream(s2):
        with torch.cuda.graph(g, stream=s2):
            eng.process_stream(x, compressed_out=comp_g, peaks_out=pk_g, stream=s2.cuda_stream)
        # (capture + instantiation leave the GPU idle for a while: the same clock ramp as in front of every timed loop --
        # without it the first replays run ~9 % slow, and so do eager launches at that moment: profiles/r03_graph_rate_probe.txt)
        clock_ramp(g.replay, torch, args.ramp_ms)
        wall, kern = timed_launches(g.replay, s2, torch, steps, warm)
    torch.cuda.synchronize()
    out["graph_replay"] = {"value": x.numel() / (wall * 1e-3), "unit": "samples/s", "ms_per_step": wall,
                           "roofline": roofline("stream_d8_f32", "stream_kernel<f32,8> (captured hipGraph, replayed)",
                                                x.numel(), byts, kern, ncu, clk)}
    out["graph_equals_eager"] = bool(torch.equal(comp, comp_g) and torch.equal(pk, pk_g))
    out["graph_over_eager"] = out["graph_replay"]["value"] / out["eager"]["value"]
    if not args.no_cpu_baseline:
        from oracle import uco
        o = uco.Oracle(uco.STREAM)
        nb = 6
        head = x[: halo + 8 * nb * hop].cpu().numpy()
        cr, pr = o.process_stream(head)
        got = comp_g[: cr.size].cpu().numpy()
        out["head_rel_err_vs_oracle"] = float(np.abs(got - cr).max() / cr.max())
        out["head_peak_offsets_equal_oracle"] = bool(np.array_equal(pk_g[:nb, 1].cpu().numpy().astype(np.uint32),
                                                                    pr["offset"][:nb]))
    eng.close()
    return out


def hello_world1(args, device, torch, mag_mean):
    """The N > 1 leg at world size 1, inside the N = 1 run: configs[4] framing, matched sweep, a uc_group of one device
    (include/uchirp.h: the RCCL communicator and the in-place all-gather of the symbol stream are made and called from C,
    on the group's gather stream, every step; three buffers in rotation as in the N > 1 run), decode.
    Gives the driver's 1 -> N efficiency a like-for-like anchor (the N = 1 contract line runs configs[1] without a gather)."""
    import uchirp
    from uchirp import synth
    nf = args.frames
    grp = uchirp.Group(uchirp.RX_REAL, devices=[device.index], mag_mean=mag_mean, time_frame=MATCHED_TIME_FRAME)
    try:
        frames, sent = synth.device_hello_frames(0, nf, device, seed=1234, snr_db=args.snr, msg=MSG)
        # a stream of its own: a NULL entry in uc_group_process_batch's stream list means "the group's own stream", and
        # torch's default stream IS the NULL stream -- the HIP events below must sit on the stream the kernel runs on
        stream = torch.cuda.Stream(device)
        gat2 = [torch.empty(nf, dtype=torch.uint8, device=device) for _ in range(NBUF)]
        torch.cuda.synchronize()

        def step(k, e0=None, e1=None):
            if e0 is not None:
                e0.record(stream)
            grp.process([frames], nf, [gat2[k % NBUF]], streams=[stream.cuda_stream])
            if e1 is not None:
                e1.record(stream)

        clock_ramp(lambda: step(0), torch, args.ramp_ms)
        for k in range(args.warmup):
            step(k)
        grp.synchronize()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(k, ev[k][0], ev[k][1])
        grp.synchronize()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        kern = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        g = gat2[(args.steps - 1) % NBUF]
        # the checker: the same frames through a plain context (uc_process_batch), no group
        eng = uchirp.Engine(uchirp.RX_REAL, device=device.index, mag_mean=mag_mean, time_frame=MATCHED_TIME_FRAME)
        ref, _ = eng.process(frames, want_stats=False)
        torch.cuda.synchronize()
        ok = bool(torch.equal(g, ref))
        eng.close()
        texts = synth.decode_hello(g.cpu().numpy(), len(MSG))
        good = sum(1 for t in texts if t == MSG)
        ms = elapsed / args.steps * 1e3
        import hashlib
        out = {"workload": "configs[4] at world size 1: %d x 2048-sample frames, K7 'Hello World!' framing, SNR %.0f dB, matched "
                           "sweep, all-gather of the symbol stream every step" % (nf, args.snr),
               "gather_backend": "uc_group_process_batch: RCCL (ncclAllGather, in place) called from C on the group's gather stream",
               "value": nf * args.steps / elapsed, "unit": "frames/s", "ms_per_step": ms, "steps": args.steps,
               "kernel_ms": kern, "gather_ms_exposed": ms - kern, "gathered_equals_decoded": ok,
               "symbols_sha256": hashlib.sha256(g.cpu().numpy().tobytes()).hexdigest(),
               "transmissions": len(texts), "transmissions_decoded_exactly": good}
        del frames, gat2
        return out
    finally:
        grp.close()


def receive_leg(args, device, torch):
    """SURVEY.md section 8 f1 as a number: the WHOLE receiver (ISR FIFO, the 8 dsp() offsets x {up, down} of every block,
    main()'s switch and resync, byte assembly: receiver/Src/main.c:417-554, 243-273, 659-668) for thousands of recorded
    microphone streams at once -- uc_receive_streams: the band kernel's ROWS build over the 8 FIFO offsets every block adds
    (ONE launch over all blocks for few streams; from 1024 streams on block by block, each step evaluating only what the switch
    can still look at -- round 6: the 4096-stream call 21.7 -> 14.6 ms) (the other 9 of its FIFO were evaluated when the block before it arrived, main.c:662; frames read through two
    base addresses from the caller's buffer, nothing packed or copied), the switch replayed on the device one wave or lane per
    stream.  Streams: 40 blocks of noise + a sample skew,
    the K7 "Hello World!" transmission rendered at 78 125 Hz, noise; generated on the device.  Real time for ONE
    microphone is 38.1 blocks/s (the MCU keeps up with exactly one)."""
    import ctypes as C
    import uchirp
    from uchirp import tx
    fs, nb = 78125.0, 176
    tone = torch.from_numpy(tx.render(MSG, fs_rx=fs, amplitude=2000.0).astype(np.float32)).to(device)
    out = {"workload": "recorded streams of %d blocks (%.2f s of microphone signal each): 40 blocks of noise + 777 samples, the K7 "
                       "'%s' transmission at 78 125 Hz (amplitude 2000, noise sigma 50), noise" % (nb, nb * N / fs, MSG),
           "blocks_per_stream": nb, "real_time_blocks_per_s_per_stream": fs / N}
    L = uchirp.lib()
    stream = torch.cuda.current_stream(device)
    # (the complex-reference receiver -- the variant whose state machine decodes the whole text; its DSP launches
    # are band_kernel<sync_cplx>, so the rocprofv3 average of the HEADLINE kernel over this command stays the contract leg's.
    # The shipping real-reference receiver runs the same call at twice the rate: profiles/r04_receive_many.txt)
    for ns, var, name in ((4096, uchirp.SYNC_CPLX, "sync_cplx_4096_streams"), (64, uchirp.SYNC_CPLX, "sync_cplx_64_streams"),
                          (1, uchirp.SYNC_CPLX, "sync_cplx_1_stream")):
        g = torch.Generator(device=device)
        g.manual_seed(ns)
        x = torch.randn((ns, nb * N), generator=g, device=device) * 50.0
        lead = 40 * N + 777
        x[:, lead:lead + tone.numel()] += tone
        eng = uchirp.Engine(var, device=device.index)
        cap = 64
        text = torch.zeros((ns, cap), dtype=torch.uint8, device=device)
        ntext = torch.zeros(ns, dtype=torch.int32, device=device)

        def call():
            rc = L.uc_receive_streams(eng._h, C.c_void_p(x.data_ptr()), uchirp.DTYPE_F32, ns, nb * N, 0, None,
                                      C.c_void_p(text.data_ptr()), cap, C.c_void_p(ntext.data_ptr()), None, 0, None,
                                      C.c_void_p(stream.cuda_stream))
            if rc != 0:
                raise RuntimeError(L.uc_last_error().decode())

        for _ in range(3):
            call()
        torch.cuda.synchronize()
        reps = 5 if ns > 64 else 20
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        texts = [bytes(r[:k]).decode("latin-1") for r, k in zip(text.cpu().numpy(), ntext.cpu().numpy())]
        out[name] = {"streams": ns, "ms_per_call": dt * 1e3, "blocks_per_s": ns * nb / dt,
                     "dsp_frames_per_s": ns * nb * 8 / dt,   # (FIFO offsets COVERED per second: from 1024 streams on not all are transformed)
                     "x_real_time": ns * nb / dt / (fs / N),
                     "streams_decoding_the_text": sum(1 for t in texts if MSG in t), "first_text": texts[0]}
        if ns == 4096:
            # the same microphones LIVE: one new block of every stream per call (uc_rx_state / uc_receive_streams_next), the
            # firmware's own mode of operation; the texts of the chunks must add up to the recorded-stream call's
            live = eng.live(ns)
            chunks = [x[:, b * N:(b + 1) * N].contiguous() for b in range(nb)]
            acc_buf, acc_len = np.zeros((ns, 4 * cap), np.uint8), np.zeros(ns, np.int64)   # what every stream has received so far
            rows_all = np.arange(ns)
            nt_host = torch.empty(ns, dtype=torch.int32).pin_memory()     # (a live host keeps pinned landing buffers)
            tx_host = torch.empty((ns, cap), dtype=torch.uint8).pin_memory()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for ch in chunks:
                rc = L.uc_receive_streams_next(eng._h, live._h, C.c_void_p(ch.data_ptr()), uchirp.DTYPE_F32, N, 0, None,
                                               C.c_void_p(text.data_ptr()), cap, C.c_void_p(ntext.data_ptr()), None, 0, None,
                                               C.c_void_p(stream.cuda_stream))
                if rc != 0:
                    raise RuntimeError(L.uc_last_error().decode())
                nt_host.copy_(ntext, non_blocking=True)        # (a live host reads its characters after every block:
                tx_host.copy_(text, non_blocking=True)         #  16 KiB of counts + 256 KiB of characters, one wait
                stream.synchronize()                           #  per 26.2 ms)
                nt = nt_host.numpy()
                if nt.any():
                    tt = tx_host.numpy()
                    for cpos in range(int(nt.max())):              # (a block completes at most one character per stream)
                        m = nt > cpos
                        acc_buf[rows_all[m], np.minimum(acc_len[m], 4 * cap - 1)] = tt[m, cpos]
                        acc_len[m] += 1
            torch.cuda.synchronize()
            dt_live = (time.perf_counter() - t0) / nb
            live.close()
            same = sum(1 for si in range(ns) if bytes(acc_buf[si, :acc_len[si]]).decode("latin-1") == texts[si])
            out["live_4096_streams"] = {"streams": ns, "blocks_per_call": 1, "calls": nb, "ms_per_call": dt_live * 1e3,
                                        "real_time_ms_per_call": N / fs * 1e3, "headroom_x_real_time": N / fs / dt_live,
                                        "microphones_served_in_real_time": int(ns * N / fs / dt_live),
                                        "streams_whose_chunks_add_up_to_the_recorded_call": same,
                                        "new_dsp_frames_per_call": ns * 8,
                                        "what": "one new block of every stream per call, the host reads the counts back after "
                                                "every call (a sync per block)"}
            # the same step with no host in the loop: back to back on one stream, and replayed from ONE captured hipGraph
            # (everything a step carries -- newest block, the 9 surviving FIFO records, main()'s locals, block counts -- lives
            # on the device); then the chain from the microphones' 1-bit PDM streams (UC_DTYPE_PDM: + the DFSDM, on the device)
            live = eng.live(ns)
            side = torch.cuda.Stream(device)
            reps_a = 60
            with torch.cuda.stream(side):
                for k in range(10):
                    live.next_into(chunks[k], text, ntext, stream=side.cuda_stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(side)
                for k in range(reps_a):
                    live.next_into(chunks[10 + k], text, ntext, stream=side.cuda_stream)
                e1.record(side)
                e1.synchronize()
                eager_ms = e0.elapsed_time(e1) / reps_a
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=side):
                    live.next_into(chunks[0], text, ntext, stream=side.cuda_stream)
                for k in range(5):
                    gr.replay()
                e0.record(side)
                for k in range(reps_a):
                    gr.replay()
                e1.record(side)
                e1.synchronize()
                graph_ms = e0.elapsed_time(e1) / reps_a
            live.reset()
            live.keep_previous(True)       # (every chunk is a buffer of its own here: uc_rx_state_keep_previous holds)
            with torch.cuda.stream(side):
                for k in range(10):
                    live.next_into(chunks[k], text, ntext, stream=side.cuda_stream)
                e0.record(side)
                for k in range(reps_a):
                    live.next_into(chunks[10 + k], text, ntext, stream=side.cuda_stream)
                e1.record(side)
                e1.synchronize()
                keep_ms = e0.elapsed_time(e1) / reps_a
            live.close()
            out["live_4096_streams"].update({"ms_per_call_back_to_back": eager_ms, "ms_per_call_graph_replay": graph_ms,
                                             "ms_per_call_back_to_back_keep_previous": keep_ms,
                                             "contracts": "ms_per_call, ..._back_to_back and ..._graph_replay: the default contract "
                                                          "(the state keeps a copy of every stream's newest block); "
                                                          "..._keep_previous: uc_rx_state_keep_previous (every chunk here is a "
                                                          "buffer of its own)",
                                             "microphones_served_in_real_time_back_to_back": int(ns * N / fs / (eager_ms * 1e-3))})
            live = eng.live(ns)
            gp = torch.Generator(device=device)
            gp.manual_seed(7)
            pdm = [torch.randint(-(1 << 31), (1 << 31) - 1, (ns, N), generator=gp, device=device, dtype=torch.int64).to(torch.int32)
                   for _ in range(4)]
            with torch.cuda.stream(side):
                for k in range(6):
                    live.next_into(pdm[k % 4], text, ntext, stream=side.cuda_stream, pdm=True)
                e0.record(side)
                for k in range(reps_a):
                    live.next_into(pdm[k % 4], text, ntext, stream=side.cuda_stream, pdm=True)
                e1.record(side)
                e1.synchronize()
            live.close()
            out["live_pdm_4096_streams"] = {"streams": ns, "ms_per_call_back_to_back": e0.elapsed_time(e1) / reps_a,
                                            "what": "one new block of every microphone per call as 2048 x 32 PDM bits "
                                                    "(UC_DTYPE_PDM, random bits: timing only; parity: tests/test_dfsdm.py): "
                                                    "sinc5 + history, ROWS band launch, replay"}
            del chunks, pdm
        eng.close()
        del x, text, ntext
    # the shipping receiver (RX_REAL) LIVE at the scale one GPU serves: 65 536 microphones, one new block each per call, back
    # to back.  Silent microphones here (noise: 94 GB would be needed for a transmission in every stream): an IDLE stream's
    # switch can look at 3 or 5 of the 8 offsets its new block adds (main.c:447-453), and only those are evaluated; a stream in
    # a tracking state costs all 8 (1.0 ms per block at this size, profiles/r05_live_async.txt)
    ns = 65536
    eng = uchirp.Engine(uchirp.RX_REAL, device=device.index)
    live = eng.live(ns)
    g = torch.Generator(device=device)
    g.manual_seed(99)
    bufs = [torch.randn((ns, N), generator=g, device=device) * 50.0 for _ in range(3)]
    text = torch.zeros((ns, 8), dtype=torch.uint8, device=device)
    ntext = torch.zeros(ns, dtype=torch.int32, device=device)
    side = torch.cuda.Stream(device)
    ms_by_contract = {}
    for contract in ("default", "keep_previous"):
        # default: the library copies every stream's newest block into the state on its way through the kernel (the caller may
        # overwrite `samples` at once); keep_previous: the caller leaves a chunk alone until the NEXT call has completed -- a ring
        # of >= 2 buffers, which this loop's three are -- and nothing is copied (uc_rx_state_keep_previous)
        live.reset()
        live.keep_previous(contract == "keep_previous")
        with torch.cuda.stream(side):
            for k in range(12):
                live.next_into(bufs[k % 3], text, ntext, stream=side.cuda_stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            for k in range(60):
                live.next_into(bufs[k % 3], text, ntext, stream=side.cuda_stream)
            e1.record(side)
            e1.synchronize()
        ms_by_contract[contract] = e0.elapsed_time(e1) / 60
    ms, ms_keep = ms_by_contract["default"], ms_by_contract["keep_previous"]
    live.close()
    eng.close()
    out["live_idle_rx_real_65536_streams"] = {"streams": ns, "ms_per_call_back_to_back": ms_keep,
                                              "contract": "uc_rx_state_keep_previous (this loop's ring of three chunk buffers keeps "
                                                          "it); the default contract -- the library copies every stream's newest "
                                                          "block into the state, what rounds 4-5 measured -- is "
                                                          "ms_per_call_back_to_back_default_contract",
                                              "ms_per_call_back_to_back_default_contract": ms,
                                              "real_time_ms_per_call": N / fs * 1e3,
                                              "microphones_served_in_real_time": int(ns * N / fs / (ms_keep * 1e-3)),
                                              "microphones_served_in_real_time_default_contract": int(ns * N / fs / (ms * 1e-3)),
                                              "what": "uc_receive_streams_next, one new block of each of 65 536 silent microphones "
                                                      "per call (IDLE streams: 3 or 5 of the 8 new FIFO offsets are evaluated, the "
                                                      "others cost nothing); keep_previous = uc_rx_state_keep_previous: the caller's "
                                                      "ring of chunk buffers is read in place, no block is copied into the state"}
    del bufs
    return out


def stream_measurement(args, eng, frames, rank, torch):
    """BASELINE config 4 (side measurement, eager launches): the batch read as ONE continuous stream through UC_STREAM."""
    x = frames.reshape(-1)
    halo, n_out, n_blocks, hop = eng.stream_geometry(x.numel())
    comp = torch.empty(n_out, dtype=torch.float32, device=x.device)
    pk = torch.empty((n_blocks, 2), dtype=torch.int32, device=x.device)
    stream = torch.cuda.current_stream(x.device)

    def launch():
        eng.process_stream(x, compressed_out=comp, peaks_out=pk, stream=stream.cuda_stream)

    clock_ramp(launch, torch, args.ramp_ms)
    wall, kern = timed_launches(launch, stream, torch, args.steps, args.warmup)
    ncu, clk = num_cus(torch, x.device), live_clock(eng, launch)
    byts = (x.numel() * 4 + n_out * 4 + n_blocks * 8) / x.numel()
    if rank == 0:
        print(json.dumps({"metric": "input samples/s (stream: FIR decimate + overlap-save compression, side measurement)",
                          "value": x.numel() / (wall * 1e-3), "unit": "samples/s", "n_gpus": 1, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": wall, "decim": int(eng.cfg.decim),
                          "blocks": n_blocks, "blocks_per_s": n_blocks / (wall * 1e-3),
                          "roofline": roofline("stream_d%d_f32" % int(eng.cfg.decim), "stream_kernel<f32,%d>" % int(eng.cfg.decim),
                                               x.numel(), byts, kern, ncu, clk)}), flush=True)


SIDE = {  # variant -> (algorithmic bytes per frame, key of profiles/r*_valu_insts.json, kernel)
    "sync_cplx": (8193, "band_sync_cplx_f32", "band_kernel<sync_cplx,f32>"),
    "compress": (8192 + 32, "compress_f32", "compress_kernel<f32>"),
    "dechirp_down": (8192 + 32, "band_dechirp_down_f32", "band_kernel<dechirp_down,f32>"),
    "iq": (8192 + 104 + 32, "iq2048_fw_f32", "iq_kernel<f32,firmware windows>"),
    "iq1024": (4096 + 104 + 32, "iq1024_fw_f32", "iq1024_kernel<f32,firmware windows>"),
    "iq_bb": (8192 + 104 + 1, "iq2048_bb_f32", "iq_kernel<f32,baseband>"),
    "iq1024_bb": (4096 + 104 + 1, "iq1024_bb_f32", "iq1024_kernel<f32,baseband>"),
}


def side_measurement(args, eng, frames, world, rank, torch):
    """Not the contract line: frames/s of one of the sibling variants on the same synthetic batch."""
    if args.variant == "stream":
        return stream_measurement(args, eng, frames, rank, torch)
    n = eng.n
    per_frame, key, kname = SIDE[args.variant]
    nfr = (frames.numel() - eng.halo - n) // n + 1
    want_sym = args.variant in ("sync_cplx", "iq_bb", "iq1024_bb")
    stats = None if want_sym else torch.empty((nfr, eng.spf, 8), dtype=torch.float32, device=frames.device)
    sym = torch.empty(nfr, dtype=torch.uint8, device=frames.device) if want_sym else None
    stream = torch.cuda.current_stream(frames.device)

    def launch():
        eng.process(frames, n_frames=nfr, want_symbols=want_sym, want_stats=not want_sym, symbols_out=sym, stats_out=stats,
                    stream=stream.cuda_stream)

    clock_ramp(launch, torch, args.ramp_ms)
    wall, kern = timed_launches(launch, stream, torch, args.steps, args.warmup)
    ncu, clk = num_cus(torch, frames.device), live_clock(eng, launch)
    if rank == 0:
        print(json.dumps({"metric": "chirp frames/s (%s, side measurement)" % args.variant, "value": nfr / (wall * 1e-3),
                          "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall,
                          "frame_len": n, "frames": nfr,
                          "roofline": roofline(key, kname, nfr, per_frame, kern, ncu, clk)}), flush=True)

