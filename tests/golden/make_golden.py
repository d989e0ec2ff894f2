#!/usr/bin/env python3
"""Regenerates tests/golden/vectors.json from the Python big-integer oracle (oracle/pasta.py).

These are ORACLE-derived vectors, not reference-derived ones: the Rust reference cannot be run
here and its tests hold no known answers.  They pin the C restatement and the HIP
path to the same canonical prime-field / prime-order-group results, and include the
surveyor-derived MinRoot known answers of SURVEY.md Appendix B.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import pasta as o  # noqa: E402

H = lambda x: "%064x" % x
out = {"note": "oracle-derived (Python big ints); canonical big-endian hex, NOT Montgomery form"}

# --- field multiplication / edge elements ---------------------------------------------------
fields = {}
for f, m in ((o.FIELD_FP, o.P), (o.FIELD_FQ, o.Q)):
    a = [0, 1, m - 1, 2, (1 << 254) % m, (m - 1) // 2] + [o.rand_fe(11, i, m) for i in range(10)]
    b = [5, m - 1, m - 1, (m + 1) // 2, (1 << 254) % m, 3] + [o.rand_fe(12, i, m) for i in range(10)]
    fields[str(f)] = {"a": [H(x) for x in a], "b": [H(x) for x in b], "mul": [H(x * y % m) for x, y in zip(a, b)]}
out["field_mul"] = fields

# --- MinRoot known answers (SURVEY.md Appendix B inputs: x=123, y=321, i=0) -------------------
mr = {}
for f in (o.FIELD_FP, o.FIELD_FQ):
    s = o.State(123, 321, 0)
    rows = {}
    for t in (1, 2, 10):
        r = o.minroot_eval(s, t, f)
        rows[str(t)] = [H(r.x), H(r.y), H(r.i)]
    mr[str(f)] = rows
out["minroot_eval_123_321_0"] = mr
r10 = o.minroot_eval(o.State(123, 321, 0), 10, o.FIELD_FQ)
out["circuit_round0_on_pallas_t10"] = [H(v) for v in o.step_witness_segment(r10, 1, o.FIELD_FQ)[:4]]

# --- step-circuit witness for t = 5 (reference test size, src/nova/proof.rs:405), i0 = 1 --------
x0 = o.rand_fe(42, 0, o.Q)
tr = o.minroot_eval_trace(o.State(x0, 0, 1), 5, o.FIELD_FQ)
out["witness_t5"] = {
    "trace_xy": [[H(s.x), H(s.y)] for s in tr], "i0": H(1),
    "W": [H(v) for v in o.step_witness_from_trace([(s.x, s.y) for s in tr], 1, 5, o.FIELD_FQ)],
}
assert out["witness_t5"]["W"] == [H(v) for v in o.step_witness_segment(tr[-1], 5, o.FIELD_FQ)]

# --- small MSMs on both curves (synthetic bases, seed 7) --------------------------------------
msm = {}
for curve in (o.CURVE_PALLAS, o.CURVE_VESTA):
    sm = o.curve_scalar_modulus(curve)
    cases = {}
    for n in (1, 2, 3, 17, 64):
        sc = [o.rand_fe(100 + n, i, sm) for i in range(n)]
        if n >= 3:
            sc[0], sc[1], sc[2] = 0, 1, sm - 1
        bases = o.synthetic_bases(curve, 7, n)
        res = o.msm_naive(sc, bases, curve)
        assert res == o.msm_by_dlog(sc, curve, 7)
        cases[str(n)] = {"scalars": [H(s) for s in sc], "bases": [[H(p[0]), H(p[1])] for p in bases],
                         "result": [H(v) for v in o.point_to_affine_ints(res)]}
    msm[str(curve)] = cases
out["msm_seed7"] = msm

# --- folding vector ops on t = 5 shape ------------------------------------------------------------
sh = o.step_circuit_shape(5, o.FIELD_FQ)
res = tr[-1]
W = [res.x, res.y, res.i] + o.step_witness_segment(res, 5, o.FIELD_FQ)
X = [res.x, res.y, res.i, tr[0].x, tr[0].y, tr[0].i]
z = W + [1] + X
az, bz, cz = o.multiply_vec(sh, z, o.Q)
assert o.is_sat_relaxed(sh, W, [0] * sh.num_cons, 1, X, o.Q)
z1 = [o.rand_fe(77, i, o.Q) for i in range(len(z))]
az1, bz1, cz1 = o.multiply_vec(sh, z1, o.Q)
u1 = z1[sh.num_vars]
T = o.cross_term(az1, bz1, cz1, az, bz, cz, u1, o.Q)
r = o.rand_fe(78, 0, o.Q) >> 126
out["fold_t5"] = {
    "shape": {"num_cons": sh.num_cons, "num_vars": sh.num_vars, "num_io": sh.num_io,
              "A": [[a, b, H(c)] for a, b, c in sh.A], "B": [[a, b, H(c)] for a, b, c in sh.B],
              "C": [[a, b, H(c)] for a, b, c in sh.C]},
    "z2": [H(v) for v in z], "az2": [H(v) for v in az], "bz2": [H(v) for v in bz], "cz2": [H(v) for v in cz],
    "z1": [H(v) for v in z1], "az1": [H(v) for v in az1], "bz1": [H(v) for v in bz1], "cz1": [H(v) for v in cz1],
    "T": [H(v) for v in T], "r": H(r), "W_fold": [H(v) for v in o.axpy(z1[:sh.num_vars], r, W, o.Q)],
}
# --- generator family 1 (seeded try-and-increment), first points on both curves -------------------------------
out["tai_bases_seed7"] = {str(c): [[H(p[0]), H(p[1])] for p in o.tai_bases(c, 7, 4)] for c in (o.CURVE_PALLAS, o.CURVE_VESTA)}

# --- compression SNARK on a folded t = 3 instance (oracle/spartan.py); same construction as tests/test_oracle_spartan.py --
from oracle import spartan as sp  # noqa: E402
t3 = 6          # 32 padded variables / constraints: one halving round of each inner-product argument, then its 16-vector
sh3 = o.step_circuit_shape(t3, o.FIELD_FQ)
def fresh(x0, i0):
    st = o.State(x0 % o.Q, 0, i0)
    rs = o.minroot_eval(st, t3, o.FIELD_FQ)
    return [rs.x, rs.y, rs.i] + o.step_witness_segment(rs, t3, o.FIELD_FQ), [rs.x, rs.y, rs.i, st.x, st.y, st.i]
W1, X1 = fresh(123456789, 5)
W2, X2 = fresh(987654321, 9)
abc1, abc2 = o.multiply_vec(sh3, W1 + [1] + X1, o.Q), o.multiply_vec(sh3, W2 + [1] + X2, o.Q)
T3 = o.cross_term(*abc1, *abc2, 1, o.Q)
r3 = 0x1234567890ABCDEF1234567890ABCDEF
Wf, Ef, uf, Xf = o.axpy(W1, r3, W2, o.Q), [r3 * v % o.Q for v in T3], (1 + r3) % o.Q, o.axpy(X1, r3, X2, o.Q)
N3 = 32
G3 = o.tai_bases(o.CURVE_PALLAS, 0x4E6F7661, N3)
U3 = o.tai_base(o.CURVE_PALLAS, 0x4E6F7661, N3)
cW3, cE3 = o.msm_naive(Wf, G3[:len(Wf)], o.CURVE_PALLAS), o.msm_naive(Ef, G3[:len(Ef)], o.CURVE_PALLAS)
pf = sp.prove(sh3, b"\x07" * 32, G3, U3, cW3, cE3, uf, Xf, Wf, Ef)
assert sp.verify(sh3, b"\x07" * 32, G3, U3, cW3, cE3, uf, Xf, pf)
fe = lambda v: int(v).to_bytes(32, "little")
pt = lambda q: b"\0" * 64 if q is None else fe(q[0]) + fe(q[1])
enc = b"".join(fe(v) for ev in pf.outer for v in ev) + b"".join(fe(v) for v in pf.claims)
enc += b"".join(fe(v) for ev in pf.inner for v in ev) + fe(pf.w_eval)
for ipa in (pf.ipa_W, pf.ipa_E):
    enc += b"".join(pt(L) + pt(R) for L, R in zip(ipa.L, ipa.R)) + b"".join(fe(v) for v in ipa.a)
out["spartan_t6"] = {"comm_W": [H(cW3[0]), H(cW3[1])], "comm_E": [H(cE3[0]), H(cE3[1])], "u": H(uf), "X": [H(v) for v in Xf],
                     "argument_hex": enc.hex()}

# --- a whole proof on the wire: Nova IVC (oracle/nova.py) + both arguments (oracle/spartan.py) + encoding (oracle/wire.py) --
import hashlib  # noqa: E402
from oracle import nova as nv, wire  # noqa: E402
wt, wn = 2, 2
winit = o.State(o.rand_fe(31, 0, o.Q), 0, 1)
wstates = [winit]
for _ in range(wn):
    wstates.append(o.minroot_eval(wstates[-1], wt, o.FIELD_FQ))
wz0 = [wstates[wn].x, wstates[wn].y, wstates[wn].i]
# the reference's step circuit (src/nova/proof.rs:155-230: 4 variables per round) and the bound form (3 per round)
for key, bound in (("wire_ivc_t2_reference", False), ("wire_ivc_t2", True)):
    wpp = nv.public_params(wt, nv.CCommit(), nv.GENS_SEED, nv.FAMILY_TRY_AND_INCREMENT, bound=bound)
    wsn = None
    for k in range(wn):
        wsn = nv.prove_step(wpp, wsn, nv.InverseMinRootCircuit(wt, wstates[wn - k], wstates[wn - k - 1], bound), wz0)
    wc = nv.compress(wpp, wsn)
    assert nv.verify_compressed(wpp, wc, wn, wz0) == ([winit.x, winit.y, winit.i], [0])
    wire_snark = wire.encode_compressed_proof(wt, wpp.params, wc)
    wire_running = wire.encode_running_proof(wt, wpp.params, wsn, wz0)
    out[key] = {"t": wt, "steps": wn, "seed": 31, "i0": 1, "bound": bound, "params": H(wpp.params),
                "compressed_proof_sha256": hashlib.sha256(wire_snark).hexdigest(), "compressed_proof_len": len(wire_snark),
                "compressed_proof_head_hex": wire_snark[:48 + 5 * 32 * 2 + 3 * 32 + 32 + 128].hex(),
                "running_proof_sha256": hashlib.sha256(wire_running).hexdigest(), "running_proof_len": len(wire_running)}
# --- the random oracle (oracle/poseidon.py): a known answer per field and the parameters digest at t = 1 ---------------
from oracle import poseidon as ps  # noqa: E402
out["ro"] = {str(f): H(ps.hash_elements(1, [1, 2, 3, 4, 5], f)) for f in (o.FIELD_FP, o.FIELD_FQ)}
out["params_t1"] = H(nv.public_params(1, None, nv.GENS_SEED, nv.FAMILY_TRY_AND_INCREMENT, bound=True).params)
out["params_t1_reference"] = H(nv.public_params(1, None, nv.GENS_SEED, nv.FAMILY_TRY_AND_INCREMENT, bound=False).params)

path = os.path.join(os.path.dirname(__file__), "vectors.json")
json.dump(out, open(path, "w"), indent=0)
print("wrote", path, os.path.getsize(path), "bytes")
