"""GPU parity of the batched per-session stage (dmz_hip_scan_sessions_batch) against the CPU oracle
(oracle/orc_session.c, itself pinned against the reference's scanner_result): bit-exact records."""
import numpy as np
import pytest

from test_oracle_vs_ref import _luhn_complete, _synthetic_session

pytestmark = pytest.mark.gpu


def _check(pkg, orc, got, want, tag):
    for name in orc.SESSION_DTYPE.names:
        if name != "reserved":
            assert np.array_equal(got[name], want[name]), (tag, name, got[name], want[name])


def test_sessions_on_synthetic_records(ctx, pkg, oracle, orc):
    rng = np.random.default_rng(404)
    S, F = 96, 20
    fr = np.zeros((S, F), orc.RESULT_DTYPE)
    ex = np.zeros((S, F), orc.EXPIRY_DTYPE)
    for s in range(S):
        n = 15 if s % 4 == 3 else 16
        prefix = [3, 7] if n == 15 else [[4], [5, 2], [6, 0, 1, 1]][s % 3]
        d = list(rng.integers(0, 10, n))
        d[: len(prefix)] = prefix
        if s % 7 != 6:
            d = _luhn_complete(d)
        mm, yy = int(rng.integers(1, 13)), int(rng.integers(24, 33))
        fr[s], ex[s] = _synthetic_session(rng, orc, F, d, [mm // 10, mm % 10, yy // 10, yy % 10],
                                          noise=0.02 + 0.2 * (s % 5 == 4))
    frp = fr.reshape(-1).view(pkg.RESULT_DTYPE)
    exp = ex.reshape(-1).view(pkg.EXPIRY_DTYPE)
    done = 0
    for scan_expiry, interval, allow_past in ((True, 0, True), (True, 100, False), (False, 33, False), (True, 33, True)):
        out = np.zeros(S, pkg.SESSION_DTYPE)
        ctx.scan_sessions(frp, exp, S, F, out, scan_expiry=scan_expiry, frame_interval_ms=interval, now_year=2026,
                          now_month=10, allow_past_expiry=allow_past)
        for s in range(S):
            want = oracle.scan_session(fr[s], ex[s], scan_expiry, interval, 2026, 10, allow_past)
            _check(pkg, orc, out[s], want, (s, scan_expiry, interval))
            done += int(want["complete"])
    assert done >= 100
    # device-resident records, no expiry records at all
    dres = ctx.alloc(frp.nbytes)
    dres.upload(frp.view(np.uint8))
    dout = ctx.alloc(S * 128)
    ctx.scan_sessions(dres.ptr, None, S, F, dout.ptr, scan_expiry=True, frame_interval_ms=200, now_year=2026, now_month=10)
    ctx.synchronize()
    out = dout.download(pkg.SESSION_DTYPE, S)
    for s in range(S):
        _check(pkg, orc, out[s], oracle.scan_session(fr[s], None, True, 200, 2026, 10, False), s)
    dres.free()
    dout.free()


def test_sessions_end_to_end_from_frames(ctx, pkg, oracle, orc):
    """sessions of repeated sightings of the same synthetic card (frame noise differs per sighting is not
    modelled: a session = F copies of one frame index plus a few other cards mixed in), pipeline -> sessions"""
    S, F = 12, 8
    idx = np.array([[1000 + s if (f % 5) else 2000 + s * F + f for f in range(F)] for s in range(S)])
    n = S * F
    y = ctx.alloc(n * pkg.FRAME_BYTES)
    for k, i in enumerate(idx.reshape(-1)):
        ctx.synth_frames(4242, int(i), 1, y.ptr + k * pkg.FRAME_BYTES)
    res = ctx.alloc(n * 1024)
    exp = ctx.alloc(n * pkg.EXPIRY_DTYPE.itemsize)
    out = ctx.alloc(S * 128)
    ctx.pipeline_expiry(y.ptr, n, res.ptr, exp.ptr)
    ctx.scan_sessions(res.ptr, exp.ptr, S, F, out.ptr, scan_expiry=True, frame_interval_ms=33, now_year=2026, now_month=10)
    ctx.synchronize()
    got = out.download(pkg.SESSION_DTYPE, S)
    gres = res.download(pkg.RESULT_DTYPE, n).reshape(S, F)
    gexp = exp.download(pkg.EXPIRY_DTYPE, n).reshape(S, F)
    numbers = 0
    for s in range(S):
        want = oracle.scan_session(gres[s].view(orc.RESULT_DTYPE), gexp[s].view(orc.EXPIRY_DTYPE), True, 33, 2026, 10, False)
        _check(pkg, orc, got[s], want, s)
        numbers += int(want["number_frame"] >= 0)
    print("sessions with an accepted number: %d of %d" % (numbers, S))
    for b in (y, res, exp, out):
        b.free()


def test_sessions_bad_arguments(ctx, pkg):
    out = np.zeros(1, pkg.SESSION_DTYPE)
    res = np.zeros(4, pkg.RESULT_DTYPE)
    with pytest.raises(pkg.DmzHipError):
        ctx.scan_sessions(None, None, 1, 4, out)
    with pytest.raises(pkg.DmzHipError):
        ctx.scan_sessions(res, None, 0, 4, out)
    with pytest.raises(pkg.DmzHipError):
        ctx.scan_sessions(res, None, 1, 4, None)
