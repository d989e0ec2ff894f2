"""CPU: the folding layer's restatement (oracle/nifs.py) -- completeness, the verifier's rejections, and the committed
wire-format vector (tests/golden/vectors.json "wire_t6": chain by oracle/nifs.py, argument by oracle/spartan.py, bytes by
oracle/wire.py).  The flow is the reference's test_nova_proof (src/nova/proof.rs:403-451) at t = 3."""
import copy
import hashlib

from oracle import nifs
from oracle import pasta as o
from oracle import spartan as sp
from oracle import wire


def _chain(t=3, n=3, seed=31, i0=1):
    init = o.State(o.rand_fe(seed, 0, o.Q), 0, i0)
    proof, sh, digest = nifs.prove_chain(init, t, n)
    st = nifs.forward_states(init, t, n)
    return init, proof, sh, digest, [st[n].x, st[n].y, st[n].i], [init.x, init.y, init.i]


def test_fold_chain_is_complete_and_satisfied():
    init, proof, sh, digest, z0, zi = _chain()
    assert len(proof.steps) == 3 and proof.steps[0].r == 0
    assert all(0 < s.r < 1 << 128 for s in proof.steps[1:])
    assert o.is_sat_relaxed(sh, proof.W, proof.E, proof.u, proof.X, o.Q)
    assert nifs.verify(proof, sh, digest, 3, z0, zi)


def test_verifier_rejections():
    init, proof, sh, digest, z0, zi = _chain()
    assert not nifs.verify(proof, sh, digest, 3, zi, zi)
    assert not nifs.verify(proof, sh, digest, 3, z0, [zi[1], zi[0], zi[2]])
    assert not nifs.verify(proof, sh, digest, 4, z0, zi)                       # wrong iterations per step
    for mutate in ("W", "E", "r", "comm_T", "X"):
        bad = copy.deepcopy(proof)
        if mutate == "W":
            bad.W[5] = (bad.W[5] + 1) % o.Q
        elif mutate == "E":
            bad.E[2] = (bad.E[2] + 1) % o.Q
        elif mutate == "r":
            bad.steps[1].r ^= 1
        elif mutate == "comm_T":
            bad.steps[2].comm_T = bad.steps[1].comm_T
        else:
            bad.steps[1].X[0] = (bad.steps[1].X[0] + 1) % o.Q
        assert not nifs.verify(bad, sh, digest, 3, z0, zi), mutate
    assert not nifs.verify(proof, sh, b"\0" * 32, 3, z0, zi)                   # other public parameters


def test_new_x_is_an_affine_image_of_another_witness_value():
    """What the product's packed commitment relies on (src/nova/proof.rs:162-173): new_x_j = y_j - (i_0 - 1 - j), so
    sum_j new_x_j G_{3+4j} = sum_j y_j G_{3+4j} - ((i_0 - 1) S0 - S1): the commitment over merged generators and the
    commitment over all of them are the same point."""
    t = 4
    init = o.State(o.rand_fe(5, 0, o.Q), 0, 9)
    res = o.minroot_eval(init, t, o.FIELD_FQ)
    W = [res.x, res.y, res.i] + o.step_witness_segment(res, t, o.FIELD_FQ)
    G = nifs.gens(len(W))
    full = nifs.commit(W)
    add = lambda a, b: o.pt_add(a, b, o.P)
    Gw = [G[0], add(G[1], G[3]), G[2]]
    for j in range(t):
        Gw += [G[4 + 4 * j], G[5 + 4 * j], add(G[6 + 4 * j], G[3 + 4 * (j + 1)]) if j + 1 < t else G[6 + 4 * j]]
    Gw.append(G[3 + 4 * t])
    Wp = W[:3] + [v for j in range(t) for v in W[4 + 4 * j:7 + 4 * j]] + [W[3 + 4 * t]]
    assert len(Wp) == 3 * t + 4 == len(Gw)
    S0 = S1 = None
    for j in range(t):
        S0, S1 = add(S0, G[3 + 4 * j]), add(S1, o.pt_mul(j, G[3 + 4 * j], o.P))
    corr = add(o.pt_mul((res.i - 1) % o.Q, S0, o.P), o.pt_neg(S1, o.P))
    packed = add(o.msm_naive(Wp, Gw, o.CURVE_PALLAS), o.pt_neg(corr, o.P))
    assert packed == full


def test_wire_golden_vector(golden):
    g = golden["wire_t6"]
    init = o.State(o.rand_fe(g["seed"], 0, o.Q), 0, g["i0"])
    proof, sh, digest = nifs.prove_chain(init, g["t"], g["steps"])
    assert digest.hex() == g["digest"]
    N = 1
    while N < max(sh.num_vars, sh.num_cons):
        N <<= 1
    arg = sp.prove(sh, digest, nifs.gens(N), nifs.gens(1, start=N)[0], nifs._pt(proof.comm_W), nifs._pt(proof.comm_E),
                   proof.u, proof.X, proof.W, proof.E)
    z = [proof.steps[0].X[:3]] + [s.X[3:] for s in proof.steps]
    cw, cT = [nifs._pt(s.comm_w) for s in proof.steps], [nifs._pt(s.comm_T) for s in proof.steps]
    blob = wire.encode_compressed_proof(g["t"], digest, z, cw, cT, arg)
    assert blob.hex() == g["compressed_proof_hex"]
    assert len(blob) == wire.chain_size(g["steps"]) + len(wire.encode_argument(arg))
    running = wire.encode_running_proof(g["t"], digest, z, cw, cT, proof.W, proof.E)
    assert len(running) == g["running_proof_len"] and hashlib.sha256(running).hexdigest() == g["running_proof_sha256"]
    # every commitment of the chain decodes back to the point it encodes
    off = 56 + 96
    for k in range(g["steps"]):
        off += 96
        assert wire.decompress_point(blob[off:off + 32]) == cw[k]
        off += 32
        if k:
            assert wire.decompress_point(blob[off:off + 32]) == cT[k]
            off += 32
