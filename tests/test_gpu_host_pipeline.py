"""-m gpu: the pipelined host-buffer path of bftkv_gpu_collective_verify (capi.hip collective_verify_pipelined).

A batch handed over in host memory is cut into pieces that are verified by private worker contexts while the pieces behind them
cross PCIe.  Items are independent, so EVERYTHING a caller can observe must be what the unsplit call gives: error bytes, exit
counts, verdict bits, fence flags, per-packet statuses with their item indices, the counters -- and, against the oracle, the
reference's verdicts.  Full size (226 MB, cut by the size rule, three calls at once on a root and two forks):
test_gpu_parity.py::test_cfg2_full_size_identity_against_the_c_oracle."""
import threading

import numpy as np
import pytest

from corpus import build as cb
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _observe(ctx, qh, tb, to, sb, so):
    err, nver, verdict = ctx.collective_verify(qh, tb, to, sb, so)
    st, st_item = ctx.last_statuses()
    return err.copy(), nver.copy(), verdict.copy(), ctx.last_fenced.copy(), st.copy(), st_item.copy(), dict(ctx.last_counters())


@pytest.mark.parametrize("early", [True, False], ids=["early-exit", "every-packet"])
def test_pieces_give_the_unsplit_answers(gpu_ctx, early):
    cl = cb.make_cluster(10, dsa_fraction=0.3)
    rates = {cb.MUT_BAD_MPI: 0.1, cb.MUT_UNKNOWN_ISSUER: 0.1, cb.MUT_DUP_SIGNER: 0.1, cb.MUT_ONE_SHORT: 0.15, cb.MUT_BAD_TAG: 0.1}
    c = cb.make_write_corpus(cl, 240, mutation_rates=rates)
    # ragged on purpose: some items without any signature stream, one with a literal-data packet (fenced), one garbage
    parts = [c.ss_data(i) for i in range(c.n_items)]
    for i in (0, 57, 58, 119, 239):
        parts[i] = b""
    parts[100] = cb.literal_packet(b"f", b"hidden") + parts[100]
    parts[101] = bytes(range(256)) * 3
    so = np.zeros(c.n_items + 1, dtype=np.uint64)
    so[1:] = np.cumsum([len(p) for p in parts], dtype=np.uint64)
    sb = np.frombuffer(b"".join(parts) + b"\0", dtype=np.uint8)[:int(so[-1])].copy()
    kr, q = H.oracle_keyring(cl), H.clique_quorum(cl)
    gpu_ctx.keyring_set(H.abi_keys(kr))
    qh = gpu_ctx.quorum_create(H.abi_qcs(q))
    gpu_ctx.set_early_exit(early)
    try:
        gpu_ctx.set_host_pipeline(1)
        base = _observe(gpu_ctx, qh, c.tbss_blob, c.tbss_off, sb, so)
        assert base[3][100] == 1 and (base[0] == 0).any() and (base[0] == 2).any()
        for pieces, copy in ((2, "ring"), (3, "direct"), (5, "ring"), (8, "direct"), (8, "ring")):
            gpu_ctx.set_host_pipeline(pieces, copy)       # both copy routes: the page-locked ring and the caller's memory directly
            got = _observe(gpu_ctx, qh, c.tbss_blob, c.tbss_off, sb, so)
            for name, a, b in zip(("err", "n_verified", "verdict", "fenced", "statuses", "status items"), base, got):
                assert np.array_equal(a, b), (pieces, copy, name)
            assert got[6] == base[6], (pieces, got[6], base[6])
            tr = gpu_ctx.host_pipeline_trace()
            assert tr["pieces"] == pieces and tr["ring"] == (copy == "ring") and tr["done_us"] > 0
        # pieces sized by a bound their streams exceed empty themselves on the device and run a second pass: same answers
        gpu_ctx.set_host_pipeline(4, "ring", tight=True)
        got = _observe(gpu_ctx, qh, c.tbss_blob, c.tbss_off, sb, so)
        for name, a, b in zip(("err", "n_verified", "verdict", "fenced", "statuses", "status items"), base, got):
            assert np.array_equal(a, b), ("second pass", name)
        assert got[6] == base[6] and gpu_ctx.host_pipeline_trace()["second_passes"] >= 3
        # and the reference's verdicts on the unfenced items
        from oracle import collective as col
        from oracle.packet import SignaturePacket
        err, nver = got[0], got[1]
        for i in range(c.n_items):
            if got[3][i]:
                continue
            r = col.collective_verify(kr, c.tbss(i), SignaturePacket(Type=1, Data=parts[i] or None), q)
            assert (err[i] == 0) == (r.err is None) and nver[i] == len(r.verified), i
        # fewer items than pieces, and a single item
        gpu_ctx.set_host_pipeline(8)
        for n in (1, 3):
            e, nv, _ = gpu_ctx.collective_verify(qh, c.tbss_blob[:int(c.tbss_off[n])], c.tbss_off[:n + 1], sb[:int(so[n])], so[:n + 1])
            assert np.array_equal(e, base[0][:n]) and np.array_equal(nv, base[1][:n])
    finally:
        gpu_ctx.set_host_pipeline(0)
        gpu_ctx.set_early_exit(True)
        gpu_ctx.quorum_destroy(qh)


def test_pipelined_calls_from_three_threads_on_a_root_and_its_forks(gpu_ctx):
    """The cgo shim's shape: several goroutines, each with its own host slices, each call pipelined over its own workers."""
    cl = cb.make_cluster(7)
    # 64 KB values: ~26 MB of payloads, i.e. ranges of several 4 MB ring slots each and slots reused within a call
    c = cb.make_write_corpus(cl, 400, value_len=65536, mutation_rates={cb.MUT_BAD_MPI: 0.1, cb.MUT_ONE_SHORT: 0.2, cb.MUT_UNKNOWN_ISSUER: 0.05})
    kr, q = H.oracle_keyring(cl), H.clique_quorum(cl)
    gpu_ctx.keyring_set(H.abi_keys(kr))
    qh = gpu_ctx.quorum_create(H.abi_qcs(q))
    gpu_ctx.set_host_pipeline(1)
    want = gpu_ctx.collective_verify(qh, c.tbss_blob, c.tbss_off, c.ss_blob, c.ss_off)
    forks = [gpu_ctx.fork() for _ in range(2)]
    ctxs = [gpu_ctx] + forks
    got, errs = {}, []

    def storm(k, cx):
        try:
            cx.set_host_pipeline(2 + k)
            got[k] = [cx.collective_verify(qh, c.tbss_blob, c.tbss_off, c.ss_blob, c.ss_off) for _ in range(4)]
        except Exception as e:      # noqa: BLE001
            errs.append((k, e))
    try:
        ths = [threading.Thread(target=storm, args=(k, cx)) for k, cx in enumerate(ctxs)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=300)
        assert not errs, errs
        assert len(got) == 3
        for res in got.values():
            for e2, nv2, vd2 in res:
                assert np.array_equal(e2, want[0]) and np.array_equal(nv2, want[1]) and np.array_equal(vd2, want[2])
    finally:
        for f in forks:
            f.close()
        gpu_ctx.set_host_pipeline(0)
        gpu_ctx.quorum_destroy(qh)


def test_pipelined_call_fails_closed(gpu_ctx):
    """An infrastructure error inside the pipelined path (a quorum handle that does not exist) is a return code, and no byte of
    the caller's result arrays reads as "verified"; non-monotone offsets are refused before anything is copied."""
    import ctypes as C
    from bftkv_amd._native import _ptr, _u8, _u64
    cl = cb.make_cluster(4)
    c = cb.make_write_corpus(cl, 64, mutation_rates={})
    gpu_ctx.keyring_set(H.abi_keys(H.oracle_keyring(cl)))
    qh = gpu_ctx.quorum_create(H.abi_qcs(H.clique_quorum(cl)))
    gpu_ctx.set_host_pipeline(4)
    try:
        tb, to, sb, so = _u8(c.tbss_blob), _u64(c.tbss_off), _u8(c.ss_blob), _u64(c.ss_off)
        n = c.n_items
        err, vd, fn = (np.zeros(n, dtype=np.uint8) for _ in range(3))
        nver = np.full(n, 7, dtype=np.uint32)
        rc = gpu_ctx.lib.bftkv_gpu_collective_verify(gpu_ctx.h, qh + 1000, n, _ptr(tb), _ptr(to), _ptr(sb), _ptr(so), _ptr(err), _ptr(nver), _ptr(vd), _ptr(fn))
        assert rc != 0 and (err != 0).all() and not vd.any() and not nver.any()
        bad = so.copy()
        bad[5] = bad[7]
        rc = gpu_ctx.lib.bftkv_gpu_collective_verify(gpu_ctx.h, qh, n, _ptr(tb), _ptr(to), _ptr(sb), _ptr(bad), _ptr(err), _ptr(nver), _ptr(vd), _ptr(fn))
        assert rc != 0
        e2, nv2, _ = gpu_ctx.collective_verify(qh, tb, to, sb, so)          # and the context is fine afterwards
        assert (e2 == 0).all() and (nv2 >= cl.suff).all()
    finally:
        gpu_ctx.set_host_pipeline(0)
        gpu_ctx.quorum_destroy(qh)


@pytest.mark.parametrize("early", [True, False], ids=["early-exit", "every-packet"])
def test_segmented_payloads_give_the_unsegmented_answers(gpu_ctx, early):
    """bftkv_gpu_collective_verify_segments: payload i = prefix_i || shared[seg_i] (TBSS ends in chunk(Cert), packet.go:192-212, the
    same certificate behind every write of a client).  Everything a caller can observe equals the call on the concatenated payloads
    -- error bytes, exit counts, verdict bits, fence flags, per-packet statuses, counters -- unsplit and cut into pieces, by both
    copy routes; with two distinct tails, payloads without a tail, empty prefixes, tails at every alignment, and text-mode / SHA-512
    signatures whose hashes run over the seam."""
    from bftkv_amd import host as HM
    cl = cb.make_cluster(10, dsa_fraction=0.3)
    rates = {cb.MUT_BAD_MPI: 0.1, cb.MUT_UNKNOWN_ISSUER: 0.1, cb.MUT_DUP_SIGNER: 0.1, cb.MUT_ONE_SHORT: 0.15, cb.MUT_BAD_TAG: 0.1}
    c = cb.make_write_corpus(cl, 240, mutation_rates=rates)
    cert = cl.client.entity
    # a second "client": the same payloads with another tail appended would not verify, so its items are built to be judged
    # anyway -- what matters is that both routes see the same bytes.  Tails of odd lengths at odd offsets; some items keep no tail.
    other = bytes(range(251)) * 5 + b"\x01\x02\x03"
    rng = np.random.default_rng(5)
    payloads, tails = [], []
    for i in range(c.n_items):
        t = c.tbss(i)
        if i % 7 == 3:
            payloads.append(t[:len(t) - len(cert)] + other); tails.append(1)      # another certificate behind the same prefix
        elif i % 11 == 5:
            payloads.append(t[:int(rng.integers(1, 200))]); tails.append(None)     # a payload that ends in neither
        elif i == 17:
            payloads.append(cert); tails.append(0)                                 # empty prefix
        else:
            payloads.append(t); tails.append(0)
    tb, to = np.frombuffer(b"".join(payloads), dtype=np.uint8), np.zeros(len(payloads) + 1, dtype=np.uint64)
    to[1:] = np.cumsum([len(p) for p in payloads], dtype=np.uint64)
    pb, po, shb, sho, seg = HM.split_tails(tb, to, [cert, other])
    assert [None if g == 0xFFFFFFFF else int(g) for g in seg] == tails and int(po[-1]) < int(to[-1]) // 3
    kr, q = H.oracle_keyring(cl), H.clique_quorum(cl)
    gpu_ctx.keyring_set(H.abi_keys(kr))
    qh = gpu_ctx.quorum_create(H.abi_qcs(q))
    gpu_ctx.set_early_exit(early)

    def observe_seg():
        err, nver, verdict = gpu_ctx.collective_verify_segments(qh, pb, po, shb, sho, seg, c.ss_blob, c.ss_off)
        st, st_item = gpu_ctx.last_statuses()
        return err.copy(), nver.copy(), verdict.copy(), gpu_ctx.last_fenced.copy(), st.copy(), st_item.copy(), dict(gpu_ctx.last_counters())
    try:
        gpu_ctx.set_host_pipeline(1)
        base = _observe(gpu_ctx, qh, tb, to, c.ss_blob, c.ss_off)
        assert (base[0] == 0).any() and (base[0] == 2).any()
        for pieces, copy in ((1, ""), (2, "ring"), (3, "direct"), (8, "direct"), (8, "ring")):
            gpu_ctx.set_host_pipeline(pieces, copy)
            got = observe_seg()
            for name, a, b in zip(("err", "n_verified", "verdict", "fenced", "statuses", "status items"), base, got):
                assert np.array_equal(a, b), (pieces, copy, name)
            assert got[6] == base[6], (pieces, got[6], base[6])
        # the reference's verdicts on the intact writes
        from oracle import collective as col
        from oracle.packet import SignaturePacket
        for i in range(c.n_items):
            if tails[i] == 0 and i != 17:
                r = col.collective_verify(kr, payloads[i], SignaturePacket(Type=1, Data=c.ss_data(i) or None), q)
                assert (got[0][i] == 0) == (r.err is None) and got[1][i] == len(r.verified), i
        # bad arguments are refused and fail closed: a segment index beyond n_shared
        bad = seg.copy(); bad[3] = 2
        with pytest.raises(Exception):
            gpu_ctx.collective_verify_segments(qh, pb, po, shb, sho, bad, c.ss_blob, c.ss_off)
    finally:
        gpu_ctx.set_host_pipeline(0)
        gpu_ctx.set_early_exit(True)
        gpu_ctx.quorum_destroy(qh)


def test_server_write_verify_sends_each_certificate_once(gpu_ctx):
    """bftkv_host_server_write_verify (Server.write's verification, server.go:286-300) hands TBSS over as prefix + certificate
    segment: the verdicts are the oracle's on the full request bytes."""
    from bftkv_amd import host as HM
    from oracle import collective as col
    from oracle.packet import SignaturePacket
    cl = cb.make_cluster(7)
    c = cb.make_write_corpus(cl, 64, mutation_rates={cb.MUT_BAD_MPI: 0.2, cb.MUT_ONE_SHORT: 0.2}, keep_requests=True)
    kr, q = H.oracle_keyring(cl), H.clique_quorum(cl)
    gpu_ctx.keyring_set(H.abi_keys(kr))
    hq = HM.Quorum.from_qcs(H.abi_qcs(q))
    err = HM.Server(gpu_ctx).write_verify(hq, c.requests)
    for i in range(c.n_items):
        r = col.collective_verify(kr, c.tbss(i), SignaturePacket(Type=1, Data=c.ss_data(i) or None), q)
        assert (err[i] == 0) == (r.err is None), i
    assert 0 < int((np.array(err) == 0).sum()) < c.n_items
