"""GPU tier: several devices behind one handle (apsu_he_multi_*, apsu_he_eval_all) — the in-process counterpart of
Receiver::RunQuery's fan-out (receiver/apsu/receiver_osn.cpp:320-364).  Runs with one device ({0}), with the same device
twice ({0, 0}: the multi-device code path — partition, per-device powers, gather — on one GPU), and with two distinct
GPUs when the box has them; every result must equal the single-context evaluation bit for bit."""
import numpy as np
import pytest

import apsu_amd
import common

pytestmark = pytest.mark.gpu


def device_sets():
    import torch
    sets = [[0], [0, 0], [0, 0, 0]]
    if torch.cuda.device_count() >= 2:
        sets += [[0, 1], [1, 0]]
    return sets


@pytest.mark.parametrize("cfg", ["toy", "1M-1024-com"])
def test_eval_all_equals_single_context(cfg):
    import torch
    if cfg == "toy":
        js, degrees = common.toy_json(), {0: [11, 10, 3, 8], 1: [7, 2, 11]}
    else:
        js, degrees = common.param_json("1M-1024-com"), {0: [124, 30], 1: [124, 5, 77]}
    S = common.make_scenario(js, degrees)
    G = apsu_amd.HeContext(js)
    rk = G.upload_relin_keys(S.rk)
    pw = G.compute_powers(S.bundle_indices, [[S.src[b][e] for e in S.sources] for b in S.bundle_indices], rk)
    gb = [G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for b in S.bundles]
    want = G.eval_bundles(gb, pw, rk, [b["mask"] for b in S.bundles])
    opw = common.oracle_powers(S)
    for i, b in enumerate(S.bundles):
        assert (want[i] == common.oracle_eval(S, opw, b)).all()
    nidx = S.p["bundle_idx_count"]
    # sources for EVERY bundle index (indices without BinBundles are never read)
    flat = []
    for b in range(nidx):
        for e in S.sources:
            flat.append(S.src[b][e] if b in S.src else S.src[S.bundle_indices[0]][e])
    units = [(b["bundle_idx"], b["cache_idx"], b["degree"]) for b in S.bundles]
    for devs in device_sets():
        M = apsu_amd.MultiContext(js, devs)
        M.upload_relin_keys(S.rk)
        slots = apsu_amd.partition_bundles(units, nidx, len(devs))
        ids = [M.upload_bundle(slots[i], b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for i, b in enumerate(S.bundles)]
        assert ids == list(range(len(S.bundles)))
        masks = [b["mask"] for b in S.bundles]
        for _ in range(2):                                        # twice: pooled buffers, steady state
            got = M.eval_all(flat, masks, G.n)
            assert (got == want).all(), devs
        # gathered onto the first device: peer copies, then ONE RCCL all-gather (falls back to peer copies when the device
        # list repeats a device, which a communicator cannot hold)
        for fl in (0, M.IO_GATHER_RCCL):
            out_d = torch.zeros((len(ids), 2, G.n), dtype=torch.int64, device="cuda:%d" % devs[0])
            M.eval_all(flat, masks, G.n, out_device_slot=0, out_ptr=out_d.data_ptr(), flags=fl)
            torch.cuda.synchronize()
            assert (out_d.cpu().numpy().view(np.uint64).reshape(want.shape) == want).all(), (devs, fl)
            assert M.last_gather() == ("rccl" if fl and len(set(devs)) == len(devs) else "peer"), (devs, fl, M.last_gather())
        # page-locked caller buffers: DMA straight from / into them
        pin_src = [apsu_amd.host_alloc(a.shape) for a in flat]
        pin_mask = [apsu_amd.host_alloc(a.shape) for a in masks]
        for d_, s_ in zip(pin_src + pin_mask, flat + masks):
            d_[...] = s_
        pin_out = apsu_amd.host_alloc((len(ids), 2, 1, G.n))
        got = M.eval_all(pin_src, pin_mask, G.n, flags=M.IO_SRC_PINNED | M.IO_MASKS_PINNED | M.IO_OUT_PINNED, out=pin_out)
        assert (got == want).all(), devs
        for a in pin_src + pin_mask + [pin_out]:
            apsu_amd.host_free(a)
        # inputs already in HBM on devices[0]: read in place there, fetched with peer copies by the others
        sd = [torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to("cuda:%d" % devs[0]) for a in flat]
        md = [torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to("cuda:%d" % devs[0]) for a in masks]
        got = M.eval_all([t.data_ptr() for t in sd], [t.data_ptr() for t in md], G.n,
                         flags=M.IO_SRC_ON_DEVICE | M.IO_MASKS_ON_DEVICE, in_device_slot=0)
        assert (got == want).all(), devs
        # phase timers: RunQuery = host wall time of the call
        M.phase_enable(True)
        M.eval_all(flat, masks, G.n)
        ph = M.phase_read()
        assert ph["Receiver::RunQuery"][0] == 1 and ph["Receiver::ComputePowers"][0] >= 1 and ph["Receiver::ProcessBinBundleCache"][0] >= 1
        assert ph["Receiver::RunQuery"][1] >= ph["Receiver::ComputePowers"][1]
        M.phase_enable(False)
        M.close()
    G.close()


@pytest.mark.parametrize("devs", [[0], [0, 0]])
def test_eval_all_reports_a_source_outside_its_prime_in_the_same_call(devs):
    """round 6 (advisor, medium): apsu_he_eval_all queues its whole query and ends with a bare stream synchronise -- a source word >= q_limb
    used to return results silently and to surface at a LATER, valid query's next wait.  Now the call that carried the bad word fails
    (peer-copy gather and the RCCL request, which falls back to peer copies for a repeated device), the next valid call succeeds with
    the right bits, and nothing is left behind for a third one"""
    js = common.param_json("1M-1024-com")
    S = common.make_scenario(js, {0: [124, 30], 1: [77]})
    nidx = S.p["bundle_idx_count"]
    flat = []
    for b in range(nidx):
        for e in S.sources:
            flat.append(S.src[b][e] if b in S.src else S.src[S.bundle_indices[0]][e])
    G = apsu_amd.HeContext(js)
    rk = G.upload_relin_keys(S.rk)
    pw = G.compute_powers(S.bundle_indices, [[S.src[b][e] for e in S.sources] for b in S.bundle_indices], rk)
    gb = [G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for b in S.bundles]
    masks = [b["mask"] for b in S.bundles]
    want = G.eval_bundles(gb, pw, rk, masks)
    units = [(b["bundle_idx"], b["cache_idx"], b["degree"]) for b in S.bundles]
    M = apsu_amd.MultiContext(js, devs)
    M.upload_relin_keys(S.rk)
    slots = apsu_amd.partition_bundles(units, nidx, len(devs))
    for i, b in enumerate(S.bundles):
        M.upload_bundle(slots[i], b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"])
    assert (M.eval_all(flat, masks, G.n) == want).all()
    wrong = [a.copy() for a in flat]
    wrong[len(S.sources) + 1][1, 0, 9] = np.uint64(int(S.C.q[0]))   # bundle index 1, second source, polynomial 1, limb 0: == q
    for fl in (0, M.IO_GATHER_RCCL):
        with pytest.raises(ValueError, match="outside"):
            M.eval_all(wrong, masks, G.n, flags=fl)
        assert (M.eval_all(flat, masks, G.n, flags=fl) == want).all()
        assert (M.eval_all(flat, masks, G.n) == want).all()
    M.close()
    G.close()


def test_multi_rejects_bad_devices_and_slots():
    js = common.toy_json()
    with pytest.raises(ValueError):
        apsu_amd.MultiContext(js, [99])
    M = apsu_amd.MultiContext(js, [0])
    with pytest.raises(ValueError):
        M.random_bundle(3, 0, 0, 5, 1)
    M.close()
