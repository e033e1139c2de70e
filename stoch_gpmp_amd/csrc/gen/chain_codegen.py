#!/usr/bin/env python3
"""Build-time code generator: compiles a serial URDF chain into straight-line, constant-folded C++
for the cost sweep (K3) -- the AOT equivalent of what a tracing compiler would do for the FK the
reference obtains from `torch_robotics` (reference cost_functions.py:51-52).

    python chain_codegen.py            # rewrites ../chain_code_generated.h for the registered chains

For each chain it emits `struct ChainCode_<name>` with
  * fk<real>(q, P): positions of the DISTINCT link frames, with the joint constants folded in
    (zero / unit entries of the joint origins vanish at generation time, the C++ compiler removes
    the rotation work no position depends on -- e.g. the whole last wrist joint of the Panda);
    the fp32 variant snaps |c| < 1e-9 to 0 (cos(1.57079632679) = 4.9e-12 is far below fp32
    resolution), the fp64 variant keeps every constant exact;
  * the link table after merging coincident frames (multiplicities) -- fields.py:79,86 sums over
    links, so k coincident links are one evaluation with weight k;
  * the list of link pairs whose distance depends on q, with weights 2 m_i m_j; rigid pairs
    contribute a constant to fields.py:124 and are folded on the host (FkPlan / selfc);
  * which distinct links never move (their sphere terms are evaluated once per wave).
Coincidence / rigidity are decided numerically from FK at random joint vectors, exactly as the
host-side analysis in api.hip does for chains that have no generated code.
"""
import math
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "robots"))
from panda_chain import PANDA_CHAIN  # noqa: E402

REGISTRY = {"panda": PANDA_CHAIN}


def rpy_matrix(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return [[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
            [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
            [-sp, cp * sr, cp * cr]]


# ------------------------------------------------------------------------------------ numeric FK
def fk_numeric(chain, q):
    R = [[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]]
    p = [0.0, 0.0, 0.0]
    out = [list(p)]
    k = 0
    for _, kind, rpy, xyz in chain:
        F = rpy_matrix(rpy)
        p = [p[r] + sum(R[r][c] * xyz[c] for c in range(3)) for r in range(3)]
        M = [[sum(R[r][k2] * F[k2][c] for k2 in range(3)) for c in range(3)] for r in range(3)]
        if kind == "revolute":
            s, c = math.sin(q[k]), math.cos(q[k])
            M = [[M[r][0] * c + M[r][1] * s, M[r][1] * c - M[r][0] * s, M[r][2]] for r in range(3)]
            k += 1
        R = M
        out.append(list(p))
    return out


def analyse(chain, probes=24):
    rnd = random.Random(1234)
    nrev = sum(1 for j in chain if j[1] == "revolute")
    L = len(chain) + 1
    d2 = [[[] for _ in range(L)] for _ in range(L)]
    moved = [0.0] * L
    first = None
    for _ in range(probes):
        q = [rnd.uniform(-3, 3) for _ in range(nrev)]
        pos = fk_numeric(chain, q)
        if first is None:
            first = pos
        for i in range(L):
            moved[i] = max(moved[i], sum((pos[i][a] - first[i][a]) ** 2 for a in range(3)))
            for j in range(i):
                d2[i][j].append(sum((pos[i][a] - pos[j][a]) ** 2 for a in range(3)))
    rep = list(range(L))
    for i in range(L):
        for j in range(i):
            if max(d2[i][j]) < 1e-20:
                rep[i] = rep[j]
                break
    reps = [i for i in range(L) if rep[i] == i]
    mult = {i: sum(1 for l in range(L) if rep[l] == i) for i in reps}
    static = {i: moved[i] < 1e-20 for i in reps}
    pairs = []
    for a, i in enumerate(reps):
        for j in reps[:a]:
            lo, hi = min(d2[i][j]), max(d2[i][j])
            if hi - lo > 1e-12 * max(1.0, hi):
                pairs.append((reps.index(i), reps.index(j), 2.0 * mult[i] * mult[j]))
    return reps, mult, static, pairs


# ------------------------------------------------------------------------------------ symbolic FK
class Sym:
    """Either a constant (`val` is a float) or a C++ expression string (`val` is a str)."""
    __slots__ = ("val",)

    def __init__(self, val):
        self.val = val

    @property
    def const(self):
        return not isinstance(self.val, str)


def lit(x):
    return repr(float(x))


def mul(a, b):
    if a.const and b.const:
        return Sym(a.val * b.val)
    if b.const:
        a, b = b, a
    if a.const:
        if a.val == 0.0:
            return Sym(0.0)
        if a.val == 1.0:
            return b
        if a.val == -1.0:
            return Sym(f"(-{b.val})")
        return Sym(f"((real){lit(a.val)} * {b.val})")
    return Sym(f"({a.val} * {b.val})")


def add(a, b):
    if a.const and b.const:
        return Sym(a.val + b.val)
    if a.const and a.val == 0.0:
        return b
    if b.const and b.val == 0.0:
        return a
    sa = f"(real){lit(a.val)}" if a.const else a.val
    sb = f"(real){lit(b.val)}" if b.const else b.val
    return Sym(f"({sa} + {sb})")


def neg(a):
    return mul(Sym(-1.0), a)


def snap(v, eps):
    if eps <= 0:
        return v
    for t in (0.0, 1.0, -1.0):
        if abs(v - t) < eps:
            return t
    return v


def gen_fk(chain, reps, eps, fname):
    lines = []
    counter = [0]

    def tmp(e):
        """Bind a non-trivial expression to a named temporary."""
        if e.const or e.val.isidentifier():
            return e
        counter[0] += 1
        name = f"v{counter[0]}"
        lines.append(f"        const real {name} = {e.val};")
        return Sym(name)

    nrev = sum(1 for j in chain if j[1] == "revolute")
    lines.append(f"    template <typename real, typename O>")
    lines.append(f"    static __device__ __forceinline__ void {fname}(const real (&q)[{nrev}], "
                 f"real (&P)[{len(reps)}][3]) {{")
    for k in range(nrev):
        lines.append(f"        real s{k}, c{k}; O::fsincos_(q[{k}], &s{k}, &c{k});")
    R = [[Sym(1.0 if r == c else 0.0) for c in range(3)] for r in range(3)]
    p = [Sym(0.0) for _ in range(3)]
    pos = [list(p)]
    k = 0
    for _, kind, rpy, xyz in chain:
        F = [[Sym(snap(v, eps)) for v in row] for row in rpy_matrix(rpy)]
        t = [Sym(snap(v, eps)) for v in xyz]
        newp = []
        for r in range(3):
            e = p[r]
            for c in range(3):
                e = add(e, mul(R[r][c], t[c]))
            newp.append(tmp(e))
        p = newp
        M = []
        for r in range(3):
            row = []
            for c in range(3):
                e = Sym(0.0)
                for k2 in range(3):
                    e = add(e, mul(R[r][k2], F[k2][c]))
                row.append(tmp(e))
            M.append(row)
        if kind == "revolute":
            s, c = Sym(f"s{k}"), Sym(f"c{k}")
            M = [[tmp(add(mul(M[r][0], c), mul(M[r][1], s))),
                  tmp(add(mul(M[r][1], c), neg(mul(M[r][0], s)))), M[r][2]] for r in range(3)]
            k += 1
        R = M
        pos.append(list(p))
    for a, l in enumerate(reps):
        for ax in range(3):
            e = pos[l][ax]
            v = f"(real){lit(e.val)}" if e.const else e.val
            lines.append(f"        P[{a}][{ax}] = {v};")
    lines.append("    }")
    return lines


def gen_chain(name, chain):
    reps, mult, static, pairs = analyse(chain)
    nrev = sum(1 for j in chain if j[1] == "revolute")
    out = [f"// chain '{name}': {len(chain)} joints, {len(chain) + 1} link frames -> {len(reps)} distinct "
           f"positions, {len(pairs)} q-dependent pairs",
           f"struct ChainCode_{name} {{",
           f"    static constexpr int N = {nrev};          // joint coordinates",
           f"    static constexpr int NJ = {len(chain)};         // joints",
           f"    static constexpr int NREP = {len(reps)};        // distinct link positions",
           f"    static constexpr int NPAIR = {len(pairs)};      // q-dependent pairs of distinct links"]

    def arr(ctype, cname, vals, fmt):
        out.append(f"    static constexpr {ctype} {cname}[{max(len(vals), 1)}] = {{"
                   + ", ".join(fmt(v) for v in vals) + "};")
    out.append("    // constexpr tables live in functions so that device code can read them as immediates")
    out.append(f"    static __host__ __device__ constexpr int rep_link(int a) {{ constexpr int t[{len(reps)}] = {{"
               + ", ".join(str(l) for l in reps) + "}; return t[a]; }")
    out.append(f"    static __host__ __device__ constexpr float mult(int a) {{ constexpr float t[{len(reps)}] = {{"
               + ", ".join(f"{mult[l]}.f" for l in reps) + "}; return t[a]; }")
    out.append(f"    static __host__ __device__ constexpr bool is_static(int a) {{ constexpr bool t[{len(reps)}] = {{"
               + ", ".join("true" if static[l] else "false" for l in reps) + "}; return t[a]; }")
    dyn = [a for a, l in enumerate(reps) if not static[l]]
    out.insert(out.index(f"    static constexpr int NPAIR = {len(pairs)};      // q-dependent pairs of distinct links") + 1,
               f"    static constexpr int NDYN = {len(dyn)};        // distinct links whose position depends on q")
    out.append(f"    static __host__ __device__ constexpr int dyn_link(int a) {{ constexpr int t[{max(len(dyn), 1)}] = {{"
               + ", ".join(map(str, dyn or [0])) + "}; return t[a]; }")
    np_ = max(len(pairs), 1)
    pi = [p[0] for p in pairs] or [0]
    pj = [p[1] for p in pairs] or [0]
    pw = [p[2] for p in pairs] or [0.0]
    out.append(f"    static __host__ __device__ constexpr int pair_i(int k) {{ constexpr int t[{np_}] = {{"
               + ", ".join(map(str, pi)) + "}; return t[k]; }")
    out.append(f"    static __host__ __device__ constexpr int pair_j(int k) {{ constexpr int t[{np_}] = {{"
               + ", ".join(map(str, pj)) + "}; return t[k]; }")
    out.append(f"    static __host__ __device__ constexpr float pair_w(int k) {{ constexpr float t[{np_}] = {{"
               + ", ".join(f"{w}f" for w in pw) + "}; return t[k]; }")
    out.append("    // joint table the runtime chain is matched against (rpy, xyz, revolute)")
    out.append(f"    static constexpr double joints[{len(chain)}][7] = {{")
    for _, kind, rpy, xyz in chain:
        out.append("        {" + ", ".join(lit(v) for v in (*rpy, *xyz)) + f", {1.0 if kind == 'revolute' else 0.0}}},")
    out.append("    };")
    out.append("    // fp32: constants within 1e-9 of 0 / +-1 snapped (below fp32 resolution)")
    out += gen_fk(chain, reps, 1e-9, "fk_snapped")
    out.append("    // exact constants (fp64 path)")
    out += gen_fk(chain, reps, 0.0, "fk_exact")
    out.append("};")
    return out


def main():
    dst = os.path.join(HERE, "..", "chain_code_generated.h")
    out = ["// GENERATED by gen/chain_codegen.py -- do not edit; re-run the generator instead.",
           "#pragma once", "#include <hip/hip_runtime.h>",
           "#pragma clang diagnostic ignored \"-Wunused-variable\"", ""]
    for name, chain in REGISTRY.items():
        out += gen_chain(name, chain)
        out.append("")
    text = "\n".join(out)
    if "--check" in sys.argv:
        ok = os.path.exists(dst) and open(dst).read() == text
        print("up to date" if ok else "STALE: re-run chain_codegen.py")
        sys.exit(0 if ok else 1)
    open(dst, "w").write(text)
    print(f"wrote {os.path.normpath(dst)} ({len(text.splitlines())} lines)")


if __name__ == "__main__":
    main()
