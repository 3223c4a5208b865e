"""N4: mask generation + block packing and the querier's decrypt/decode on the GPU, against the oracle
(receiver_osn.cpp:53-73,217-284 ; result_package.cpp:175-213 ; sender_osn.cpp:675-700)."""
import numpy as np
import pytest

import common
from common import ref

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import apsu_amd                                            # noqa: E402
from bench import splitmix_values                          # noqa: E402


def _masks(G, C, js_felts, seed, count):
    n = G.n
    buf = torch.empty(count * n, dtype=torch.int64, device="cuda")
    vals, blks = G.mask_generate(seed, count, buf.data_ptr())
    torch.cuda.synchronize()
    enc = buf.cpu().numpy().view(np.uint64).reshape(count, n)
    for c in range(count):
        assert (vals[c] == splitmix_values(seed, c, n, G.t)).all()          # the documented generator
        assert (vals[c] < G.t).all()
        assert (enc[c] == C.encode(vals[c])).all()                           # BatchEncoder::encode
        assert (blks[c] == C.vec_to_oc_block(vals[c], js_felts)[: blks.shape[1]]).all()
    return buf, vals, blks, enc


@pytest.mark.parametrize("felts", [5, 6, 8])
def test_mask_generate_toy(felts):
    # even and odd felts_per_item: the odd branch of vec_to_oc_block has its own shift
    js = common.toy_json(felts=felts)
    C = ref.RefContext.from_params(ref.load_params(js))
    G = apsu_amd.HeContext(js)
    _, vals, blks, _ = _masks(G, C, felts, 0x1234, 3)
    assert blks.shape == (3, G.n // felts, 2)
    assert len({int(v) for v in vals[0][:64]}) > 32                           # not constant


def test_mask_generate_16M_params():
    js = common.param_json("16M-4096")
    C = ref.RefContext.from_params(ref.load_params(js))
    G = apsu_amd.HeContext(js)
    _masks(G, C, 5, 77, 2)


@pytest.mark.parametrize("cfg", ["toy", "16M-4096", "1M-1024-com"])
def test_mask_generate_blake2xb_stream(cfg):
    """the reference's generator (receiver_osn.cpp:221-224,248-251): SEAL's Blake2xb PRNG under a 64-byte seed, 32-bit draws
    % plain_modulus, against the Python model of it (oracle/blake2x.py); encode and packing as for the other generator;
    a second call continues the stream where the first stopped"""
    from oracle import blake2x
    js = common.toy_json(felts=5) if cfg == "toy" else common.param_json(cfg)
    C = ref.RefContext.from_params(ref.load_params(js))
    G = apsu_amd.HeContext(js)
    felts = ref.load_params(js)["felts_per_item"]
    n, count = G.n, 3
    seed = [(0x0123456789abcdef * (i + 3)) & ((1 << 64) - 1) for i in range(8)]
    model = np.array(blake2x.Blake2xbPRNG(seed).values(max((count + 1) * n, 1021 + n)), dtype=np.uint64) % np.uint64(G.t)
    buf = torch.empty((count + 1) * n, dtype=torch.int64, device="cuda")
    vals, blks = G.mask_generate_blake2xb(seed, count, buf.data_ptr())
    vals2, _ = G.mask_generate_blake2xb(seed, 1, buf.data_ptr() + count * n * 8, first_value=count * n)
    torch.cuda.synchronize()
    enc = buf.cpu().numpy().view(np.uint64).reshape(count + 1, n)
    assert (vals.reshape(-1) == model[: count * n]).all()
    assert (vals2.reshape(-1) == model[count * n:(count + 1) * n]).all()
    for c in range(count):
        assert (enc[c] == C.encode(vals[c])).all()
        assert (blks[c] == C.vec_to_oc_block(vals[c], felts)[: blks.shape[1]]).all()
    assert (enc[count] == C.encode(vals2[0])).all()
    # an unaligned continuation (not a multiple of the 16 values of a BLAKE2b block)
    v3, _ = G.mask_generate_blake2xb(seed, 1, buf.data_ptr(), first_value=1021, want_blocks=False)
    assert (v3[0] == model[1021:1021 + n]).all()
    with pytest.raises(ValueError):
        G.mask_generate_blake2xb(seed[:7], 1, buf.data_ptr())
    G.close()


def test_loopback_masks_eval_decrypt():
    """all-GPU loopback: masks drawn on the device, evaluation with device-resident masks, the querier's
    decrypt/decode/packing on the device; every stage equals the oracle and the plaintext meaning holds"""
    js = common.toy_json()
    S = common.make_scenario(js, {0: [11, 4, 0], 1: [7]})
    G = apsu_amd.HeContext(js)
    n = G.n
    count = len(S.bundles)
    buf, vals, blks, enc = _masks(G, S.C, 5, 99, count)          # toy_json: felts_per_item = 5
    for i, b in enumerate(S.bundles):                                        # the scenario now uses the GPU's masks
        b["mask_vals"], b["mask"] = vals[i], enc[i]
    rk = G.upload_relin_keys(S.rk) if S.rk is not None else None
    pw = G.compute_powers(S.bundle_indices, [[S.src[b][e] for e in S.sources] for b in S.bundle_indices], rk)
    gb = [G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for b in S.bundles]
    out = G.eval_bundles(gb, pw, rk, [buf.data_ptr() + i * n * 8 for i in range(count)], masks_on_device=True)
    opw = common.oracle_powers(S)
    for i, b in enumerate(S.bundles):
        assert (out[i] == common.oracle_eval(S, opw, b)).all()
    got, gblk = G.decrypt_decode(S.sk[0], out)
    felts = 5
    for i, b in enumerate(S.bundles):
        pt, budget = S.C.decrypt(S.sk, np.ascontiguousarray(out[i]), 0)
        assert budget > 0
        exp = S.C.decode(pt)
        assert (got[i] == exp).all()                                          # Decryptor::decrypt + BatchEncoder::decode
        assert (got[i].astype(object) == common.expected_slots(S, b)).all()   # P(x) + mask
        assert (gblk[i] == S.C.vec_to_oc_block(exp, felts)[: gblk.shape[1]]).all()
    # device-resident ciphertexts give the same answer
    dct = torch.from_numpy(np.ascontiguousarray(out).view(np.int64)).cuda()
    got2, _ = G.decrypt_decode(S.sk[0], dct.data_ptr(), count=count, on_device=True, want_blocks=False)
    assert (got2 == got).all()


def test_n4_rejects_bad_input():
    js = common.toy_json()
    G = apsu_amd.HeContext(js)
    with pytest.raises(ValueError):                                       # secret key not reduced modulo q_0
        G.decrypt_decode(np.full(G.n, G.q[0], dtype=np.uint64), np.zeros((1, 2, 1, G.n), dtype=np.uint64))
    vals, blks = G.mask_generate(1, 0, 0)                                 # count = 0: nothing to do
    assert vals.shape[0] == 0 and blks.shape[0] == 0


def test_all_gpu_loopback_1M_params():
    """N1 + hot path + N4 chained on the device at 1M-1024-com: BinBundle built from items on the GPU, masks drawn on the
    GPU, evaluation, the querier's decrypt/decode on the GPU.  Only key generation and query encryption (the other
    party) come from the oracle.  A slot decrypts to its mask exactly where the query value is an item of the bin."""
    rng = np.random.default_rng(21)
    js = common.param_json("1M-1024-com")
    S = common.make_scenario(js, {0: []})
    G = apsu_amd.HeContext(js)
    n = G.n
    n_bins = S.p["items_per_bundle"] * S.p["felts_per_item"]
    bins = []
    for s in range(n_bins):
        c = int(rng.integers(1, 100))
        b = [int(v) for v in rng.choice(G.t - 1, size=c, replace=False) + 1]
        if s % 3 == 0:
            b[int(rng.integers(0, c))] = int(S.x[0][s])          # the query value is a member of every third bin
        bins.append(b)
    gb = G.build_bundle(0, 0, bins)
    buf = torch.empty(n, dtype=torch.int64, device="cuda")
    mask_vals, blocks = G.mask_generate(2024, 1, buf.data_ptr())
    rk = G.upload_relin_keys(S.rk)
    pw = G.compute_powers([0], [[S.src[0][e] for e in S.sources]], rk)
    out = torch.empty((1, 2, n), dtype=torch.int64, device="cuda")
    G.eval_bundles([gb], pw, rk, [buf.data_ptr()], out=out.data_ptr(), masks_on_device=True, out_on_device=True)
    got, gblk = G.decrypt_decode(S.sk[0], out.data_ptr(), count=1, on_device=True)
    member = np.array([int(S.x[0][s]) in bins[s] for s in range(n_bins)])
    assert member[::3].all()
    assert (got[0][:n_bins][member] == mask_vals[0][:n_bins][member]).all()
    assert (got[0][:n_bins][~member] != mask_vals[0][:n_bins][~member]).mean() > 0.99       # P(x) != 0 off the set
    # a matching item gives the same PEQT block on both sides
    felts = S.p["felts_per_item"]
    items_all_member = [i for i in range(S.p["items_per_bundle"]) if member[i * felts:(i + 1) * felts].all()]
    for i in items_all_member:
        assert (gblk[0][i] == blocks[0][i]).all()
    # and the host-side oracle decrypts the same plaintext
    pt, budget = S.C.decrypt(S.sk, np.ascontiguousarray(out.cpu().numpy().view(np.uint64).reshape(2, 1, n)), 0)
    assert budget > 0 and (S.C.decode(pt) == got[0]).all()
