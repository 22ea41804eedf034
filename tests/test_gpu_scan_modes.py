"""dmz_hip_scan_cards_batch mode bits: DMZ_HIP_SCAN_SKIP_NUMBER = scan_card_image(collect_card_number = false)
(frame.cpp:43-49, scan.cpp:43-48: what the reference runs once a session's number is accepted) and
DMZ_HIP_SCAN_ONLY_WARPED on records that carry stale gate flags."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 0xCA4D10


def test_skip_number_mode_marks_vseg_ok_frames_usable_and_categorises_their_expiry(ctx, pkg, oracle):
    n = 96
    cards = ctx.alloc(n * pkg.CARD_BYTES)
    ctx.synth_cards(SEED, 500, n, cards.ptr)
    host_cards = cards.download(np.uint8).reshape(n, 270, 428).copy()
    host_cards[5] = 0                         # fails the vseg gate
    host_cards[6] = host_cards[6][::-1, ::-1]  # upside down
    cards.upload(host_cards)
    full = np.zeros(n, pkg.RESULT_DTYPE)
    ctx.scan_cards(cards.ptr, n, full)
    got = np.zeros(n, pkg.RESULT_DTYPE)
    got["flags"] = 0x7  # stale bits from an earlier use of the buffer must not survive
    got["n_offsets"] = 9
    ctx.scan_cards(cards.ptr, n, got, skip_number=True)
    exp = np.zeros(n, pkg.EXPIRY_DTYPE)
    ctx.scan_expiry(cards.ptr, n, got, exp)
    promoted = 0
    for i in range(n):
        want = oracle.scan_card_image(host_cards[i], warped=False, collect_card_number=False)
        g = got[i]
        assert g["flags"] == want["flags"], i
        assert g["vseg_y_offset"] == want["vseg_y_offset"] and g["pattern_type"] == want["pattern_type"], i
        assert abs(float(g["vseg_score"]) - float(want["vseg_score"])) <= 1e-4
        assert g["n_offsets"] == 0 and not g["offsets"].any() and not g["scores"].any() and g["number_score"] == 0
        vseg_ok = bool(want["flags"] & pkg.FLAG_VSEG_OK)
        assert bool(g["flags"] & pkg.FLAG_USABLE) == vseg_ok  # frame.cpp:43-49
        promoted += int(vseg_ok and not (full[i]["flags"] & pkg.FLAG_USABLE))
        we = oracle.scan_card_expiry(host_cards[i], want)
        ge = exp[i]
        assert ge["n_found"] == we["n_found"] and ge["categorised"] == we["categorised"], i
        if vseg_ok and want["vseg_y_offset"] < 240:
            assert ge["categorised"] == 1, i
        k = int(we["n_groups"])
        assert np.array_equal(ge["groups"]["char_left"][:k], we["groups"]["char_left"][:k]), i
        if k:
            assert np.abs(ge["groups"]["scores"][:k] - we["groups"]["scores"][:k]).max() <= 1e-4
    # the case the mode exists for: frames whose digit score fails the number gate still count for the expiry
    assert promoted >= 5
    assert not (got[5]["flags"] & pkg.FLAG_VSEG_OK) and (got[6]["flags"] & pkg.FLAG_UPSIDE_DOWN)
    cards.free()


def test_only_warped_clears_stale_gate_flags(ctx, pkg):
    n = 8
    cards = ctx.alloc(n * pkg.CARD_BYTES)
    ctx.synth_cards(SEED, 0, n, cards.ptr)
    rec = np.zeros(n, pkg.RESULT_DTYPE)
    rec["flags"] = pkg.FLAG_VSEG_OK | pkg.FLAG_USABLE  # stale: not WARPED
    rec["flags"][::2] |= pkg.FLAG_WARPED
    rec["n_offsets"] = 16
    rec["scores"] = 0.5
    ctx.scan_cards(cards.ptr, n, rec, only_warped=True)
    exp = np.zeros(n, pkg.EXPIRY_DTYPE)
    ctx.scan_expiry(cards.ptr, n, rec, exp)
    for i in range(n):
        if i % 2 == 0:
            assert rec[i]["flags"] & pkg.FLAG_WARPED and rec[i]["flags"] & pkg.FLAG_VSEG_OK
        else:
            assert rec[i]["flags"] == 0 and rec[i]["n_offsets"] == 0 and not rec[i]["scores"].any()
            assert exp[i]["n_found"] == 0 and exp[i]["categorised"] == 0
    with pytest.raises(pkg.DmzHipError):
        ctx.lib.dmz_hip_scan_cards_batch.restype  # noqa: B018 (attribute exists)
        ctx._check(ctx.lib.dmz_hip_scan_cards_batch(ctx.h, cards.ptr, pkg.CARD_BYTES, n, 4, rec.ctypes.data))
    cards.free()


def test_two_queue_schedule_gives_the_same_bytes_as_one_queue(ctx, pkg):
    """dmz_hip_set_two_queues: the expiry segmentation beside hseg + digits on a second device queue, joined before
    the expiry CNN -- every record, expiry record and card byte equal to the single-queue run."""
    n = 3072
    y = ctx.alloc(n * pkg.FRAME_BYTES)
    ctx.synth_frames(SEED, 9000, n, y.ptr)
    outs = []
    for two in (True, False, True):
        ctx.set_two_queues(two)
        res = ctx.alloc(n * 1024)
        cards = ctx.alloc(n * pkg.CARD_BYTES)
        exp = ctx.alloc(n * pkg.EXPIRY_DTYPE.itemsize)
        ctx.pipeline_expiry(y.ptr, n, res.ptr, exp.ptr, cards.ptr)
        ctx.synchronize()
        outs.append((res.download(np.uint8).copy(), exp.download(np.uint8).copy(), cards.download(np.uint8).copy()))
    ctx.set_two_queues(True)
    for k in (1, 2):
        for a, b in zip(outs[0], outs[k]):
            assert np.array_equal(a, b)
    got = outs[0][0].view(pkg.RESULT_DTYPE)
    assert (got["flags"] & pkg.FLAG_USABLE).astype(bool).sum() > n // 2  # the batch does exercise the expiry CNN


def test_matrix_core_loops_are_run_to_run_identical(ctx, pkg):
    """Gate for the unfenced load -> matrix-instruction loops (expiry conv1, the digit convolution and its chunked FC1):
    a batch large enough to keep several workgroups resident on every CU is run repeatedly under every schedule that
    changes which kernels share a CU (one queue / three queues) and with every expiry conv variant; all records must be
    the same BYTES from run to run within a variant (an operand hazard between co-resident workgroups showed up in round 2
    as accumulators that differed from run to run; the 1e-4 parity tolerance would not see it)."""
    n = 8192
    y = ctx.alloc(n * pkg.FRAME_BYTES)
    ctx.synth_frames(SEED, 20000, n, y.ptr)
    res = ctx.alloc(n * 1024)
    cards = ctx.alloc(n * pkg.CARD_BYTES)
    exp = ctx.alloc(n * pkg.EXPIRY_DTYPE.itemsize)
    try:
        for mode in (pkg.EXPIRY_CONV_F16X3, pkg.EXPIRY_CONV_BF16X3, pkg.EXPIRY_CONV_F32, pkg.EXPIRY_CONV_BF16):
            ctx.set_expiry_conv(mode)
            first = None
            for run, two in enumerate((True, False, True, False)):
                ctx.set_two_queues(two)
                ctx.pipeline_expiry(y.ptr, n, res.ptr, exp.ptr, cards.ptr)
                ctx.synchronize()
                got = (res.download(np.uint8).copy(), exp.download(np.uint8).copy())
                if first is None:
                    first = got
                    e = got[1].view(pkg.EXPIRY_DTYPE)
                    assert (e["categorised"] > 0).sum() > n // 4 and (e["n_groups"] > 0).sum() > n // 4
                else:
                    assert np.array_equal(first[0], got[0]), ("frame records", mode, run)
                    assert np.array_equal(first[1], got[1]), ("expiry records", mode, run)
    finally:
        ctx.set_expiry_conv(pkg.EXPIRY_CONV_F16X3)
        ctx.set_two_queues(True)
        for b in (y, res, cards, exp):
            b.free()


def test_capi_gather_at_world_size_one(ctx, pkg):
    """The C-ABI's multi-GPU entry points on the one GPU a test box has: communicator of one rank (with an RCCL id when
    librccl loads, and the RCCL-free form), the asynchronous gather of both record types behind a pipeline call, two
    alternating destinations, wait.  The N > 1 transfers (ncclSend / ncclRecv into the root) cannot run here: they are
    covered by construction (the same shard ranges, tests/test_capi_exports.py) and by the driver's multi-GPU run."""
    n = 1024
    y = ctx.alloc(n * pkg.FRAME_BYTES)
    ctx.synth_frames(SEED, 31000, n, y.ptr)
    res, exp, cards = ctx.alloc(n * 1024), ctx.alloc(n * pkg.EXPIRY_DTYPE.itemsize), ctx.alloc(n * pkg.CARD_BYTES)
    dst = [(ctx.alloc(n * 1024), ctx.alloc(n * pkg.EXPIRY_DTYPE.itemsize)) for _ in range(2)]
    try:
        uid = pkg.comm_unique_id()
    except pkg.DmzHipError:
        uid = None
    for use_id in ([True, False] if uid is not None else [False]):
        ctx.comm_init(1, 0, uid if use_id else None)
        try:
            for step in range(3):
                slot = step & 1
                ctx.pipeline_expiry(y.ptr, n, res.ptr, exp.ptr, cards.ptr)
                ctx.gather_wait(2 * slot, host_sync=False)      # the gathers that used this pair of destinations two steps ago
                ctx.gather_wait(2 * slot + 1, host_sync=False)
                ctx.gather_records(res.ptr, 1024, n, 0, dst[slot][0].ptr, slot=2 * slot)
                ctx.gather_records(exp.ptr, pkg.EXPIRY_DTYPE.itemsize, n, 0, dst[slot][1].ptr, slot=2 * slot + 1)
                ctx.gather_wait(-1, host_sync=True)
                assert np.array_equal(dst[slot][0].download(np.uint8), res.download(np.uint8))
                assert np.array_equal(dst[slot][1].download(np.uint8), exp.download(np.uint8))
            with pytest.raises(pkg.DmzHipError):
                ctx.gather_records(res.ptr, 1024, n, 1, dst[0][0].ptr)  # root outside the communicator
            with pytest.raises(pkg.DmzHipError):
                ctx.gather_records(res.ptr, 1024, n, 0, dst[0][0].ptr, slot=99)
        finally:
            ctx.comm_destroy()
    with pytest.raises(pkg.DmzHipError):
        ctx.gather_records(res.ptr, 1024, n, 0, dst[0][0].ptr)  # no communicator
    for b in (y, res, exp, cards, dst[0][0], dst[0][1], dst[1][0], dst[1][1]):
        b.free()


def test_two_contexts_driven_from_two_host_threads(pkg, ctx):
    """one process OR ONE THREAD per GPU.  Two contexts (here on the one GPU a test box has), each
    driven from its own host thread at the same time: context creation of the second one inside its thread, the whole
    pipeline, communicator of one rank (librccl is loaded lazily -- both threads reach the loader together, as the collective
    ncclCommInitRank makes every thread of a multi-GPU host do) and the gather of both record types.  Every thread's bytes
    must equal what one thread alone produces."""
    import threading
    n, rounds = 1536, 3
    frames = ctx.alloc(n * pkg.FRAME_BYTES)
    ctx.synth_frames(SEED, 52000, n, frames.ptr)
    host_frames = frames.download(np.uint8)
    want_res = np.zeros(n, pkg.RESULT_DTYPE)
    want_exp = np.zeros(n, pkg.EXPIRY_DTYPE)
    ctx.pipeline_expiry(frames.ptr, n, want_res, want_exp)
    frames.free()
    try:
        uid_ok = pkg.comm_unique_id() is not None
    except pkg.DmzHipError:
        uid_ok = False
    out, errors = {}, []
    start = threading.Barrier(2)

    def worker(t):
        try:
            start.wait()
            c = pkg.Context(0)
            try:
                y = c.alloc(n * pkg.FRAME_BYTES).upload(host_frames)
                res, exp, cards = c.alloc(n * 1024), c.alloc(n * pkg.EXPIRY_DTYPE.itemsize), c.alloc(n * pkg.CARD_BYTES)
                dst = (c.alloc(n * 1024), c.alloc(n * pkg.EXPIRY_DTYPE.itemsize))
                start.wait()
                c.comm_init(1, 0, pkg.comm_unique_id() if uid_ok else None)
                got = []
                for _ in range(rounds):
                    c.pipeline_expiry(y.ptr, n, res.ptr, exp.ptr, cards.ptr)
                    c.gather_records(res.ptr, 1024, n, 0, dst[0].ptr, slot=0)
                    c.gather_records(exp.ptr, pkg.EXPIRY_DTYPE.itemsize, n, 0, dst[1].ptr, slot=1)
                    c.gather_wait(-1, host_sync=True)
                    got.append((dst[0].download(np.uint8).copy(), dst[1].download(np.uint8).copy()))
                c.comm_destroy()
                out[t] = got
                for b in (y, res, exp, cards, dst[0], dst[1]):
                    b.free()
            finally:
                c.close()
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((t, repr(e)))
            try:
                start.abort()
            except Exception:  # noqa: BLE001
                pass

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(300)
    assert not errors, errors
    for t in range(2):
        for r, x in out[t]:
            assert r.tobytes() == want_res.tobytes(), t
            assert x.tobytes() == want_exp.tobytes(), t
