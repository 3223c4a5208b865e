"""Independent Python big-integer model of the BFV operations on APSU's DB-side hot path.

ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (Microsoft SEAL is absent from
/root/reference and this image).  This is the SECOND, independent statement of
SURVEY.md App. B used to pin the C oracle (oracle/ref_*.c): it shares no code with it and is
written at the mathematical level (Python ints, CRT, schoolbook negacyclic products, direct
polynomial evaluation for the NTT) so that an indexing / table / laziness mistake in the C
oracle or in the HIP kernels cannot be mirrored here.

Reference call sites being modelled: receiver/apsu/receiver_osn.cpp:395-488,
receiver/apsu/bin_bundle.cpp:67-174,192-360.  SEAL definitions: [SEAL-recall], App. B1-B10.
"""
from functools import reduce


# ----------------------------------------------------------------------------- numbers (B1)
def is_prime(v):
    if v < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if v % p == 0:
            return v == p
    d, r = v - 1, 0
    while d % 2 == 0:
        d //= 2
        r += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, v)
        if x in (1, v - 1):
            continue
        for _ in range(r - 1):
            x = x * x % v
            if x == v - 1:
                break
        else:
            return False
    return True


def get_primes(factor, bits, count):
    out, v = [], ((1 << bits) - 1) // factor * factor + 1
    while len(out) < count and v > (1 << (bits - 1)):
        if is_prime(v):
            out.append(v)
        v -= factor
    assert len(out) == count
    return out


def coeff_modulus_create(n, bit_sizes):
    lists = {b: get_primes(2 * n, b, bit_sizes.count(b)) for b in set(bit_sizes)}
    return [lists[b].pop() for b in bit_sizes]


def minimal_primitive_root(m, q):
    """smallest primitive m-th root of unity mod q (m power of two)."""
    g = 2
    while True:
        r = pow(g, (q - 1) // m, q)
        if pow(r, m // 2, q) == q - 1:
            break
        g += 1
    return min(pow(r, k, q) for k in range(1, m, 2))


def brv(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


def prod(xs):
    return reduce(lambda a, b: a * b, xs, 1)


# ----------------------------------------------------------------------------- context
class Model:
    def __init__(self, n, coeff_modulus, t):
        self.n, self.logn = n, n.bit_length() - 1
        self.key_q = list(coeff_modulus)
        self.K = len(coeff_modulus)
        self.t = t
        self.first = self.K - 2 if self.K > 1 else 0
        self.psi = {q: minimal_primitive_root(2 * n, q) for q in self.key_q}
        maxL = self.first + 1
        self.bcp = get_primes(2 * n, 61, maxL + 3)
        for p in self.bcp:
            self.psi[p] = minimal_primitive_root(2 * n, p)
        if (t - 1) % (2 * n) == 0:
            self.psi[t] = minimal_primitive_root(2 * n, t)
        self.m_tilde = 1 << 32

    @classmethod
    def from_bits(cls, n, coeff_bits, plain_modulus=0, plain_bits=0):
        q = coeff_modulus_create(n, list(coeff_bits))
        t = plain_modulus or coeff_modulus_create(n, [plain_bits])[0]
        return cls(n, q, t)

    def base(self, chain_idx):
        return self.key_q[: chain_idx + 1]

    def clamp(self, chain_idx):
        return min(chain_idx, self.first)

    def rns_tool(self, chain_idx):
        q = self.base(chain_idx)
        L = len(q)
        Q = prod(q)
        nB = L + (1 if 32 + self.t.bit_length() + Q.bit_length() >= 61 * L + 61 else 0)
        m_sk, gamma = self.bcp[0], self.bcp[1]
        B = self.bcp[2: 2 + nB]
        return q, Q, B, m_sk, gamma

    # ------------------------------------------------------------------------- NTT (B3)
    def ntt(self, a, q):
        """out[i] = a(psi^(2*brv(i)+1)) by direct evaluation (O(n^2))."""
        n, psi = self.n, self.psi[q]
        out = []
        for i in range(n):
            x = pow(psi, 2 * brv(i, self.logn) + 1, q)
            acc = 0
            for c in reversed(a):
                acc = (acc * x + c) % q
            out.append(acc)
        return out

    def intt(self, A, q):
        """inverse of ntt() by the orthogonality relation."""
        n, psi = self.n, self.psi[q]
        ninv = pow(n, -1, q)
        ipsi = pow(psi, -1, q)
        xs = [pow(ipsi, 2 * brv(i, self.logn) + 1, q) for i in range(n)]
        out = []
        for j in range(n):
            acc = 0
            for i in range(n):
                acc += A[i] * pow(xs[i], j, q)
            out.append(acc % q * ninv % q)
        return out

    def negacyclic_mul(self, a, b, q):
        n = self.n
        out = [0] * n
        for i, x in enumerate(a):
            if not x:
                continue
            for j, y in enumerate(b):
                k = i + j
                if k < n:
                    out[k] += x * y
                else:
                    out[k - n] -= x * y
        return [v % q for v in out]

    # ------------------------------------------------------------------------- helpers
    # a ciphertext is a list (polys) of lists (limbs) of lists (n coefficients)
    def crt(self, limbs, q):
        Q = prod(q)
        out = []
        for k in range(self.n):
            x = 0
            for j, qj in enumerate(q):
                Qj = Q // qj
                x += limbs[j][k] * pow(Qj, -1, qj) % qj * Qj
            out.append(x % Q)
        return out

    # ------------------------------------------------------------------------- ops
    def transform_to_ntt(self, ct, chain_idx):
        return [[self.ntt(l, q) for l, q in zip(p, self.base(chain_idx))] for p in ct]

    def transform_from_ntt(self, ct, chain_idx):
        return [[self.intt(l, q) for l, q in zip(p, self.base(chain_idx))] for p in ct]

    def multiply_plain_ntt(self, ct, pt_ntt, chain_idx):
        return [[[x * y % q for x, y in zip(l, pl)] for l, pl, q in zip(p, pt_ntt, self.base(chain_idx))] for p in ct]

    def plain_lift(self, pt, chain_idx):            # B5
        th = (self.t + 1) // 2
        return [[(c + (q - self.t) if c >= th else c) for c in pt] for q in self.base(chain_idx)]

    def plain_lift_ntt(self, pt, chain_idx):
        return [self.ntt(l, q) for l, q in zip(self.plain_lift(pt, chain_idx), self.base(chain_idx))]

    def multiply_plain_coeff(self, ct, pt, chain_idx):   # B6 via schoolbook product (no NTT)
        nz = [k for k, c in enumerate(pt) if c]
        if len(nz) == 1:                                  # monomial shortcut: no lift [SEAL-recall]
            lifted = [list(pt) for _ in self.base(chain_idx)]
        else:
            lifted = self.plain_lift(pt, chain_idx)
        return [[self.negacyclic_mul(l, pl, q) for l, pl, q in zip(p, lifted, self.base(chain_idx))] for p in ct]

    def add(self, a, b, chain_idx):
        """Evaluator::add_inplace: operands may differ in size, the longer one's extra polynomials are copied"""
        if len(a) < len(b):
            a, b = b, a
        head = [[[(x + y) % q for x, y in zip(la, lb)] for la, lb, q in zip(pa, pb, self.base(chain_idx))]
                for pa, pb in zip(a, b)]
        return head + [[list(l) for l in p] for p in a[len(b):]]

    def add_plain(self, ct, pt, chain_idx):          # B7, big-int statement
        q = self.base(chain_idx)
        Q, t = prod(q), self.t
        scaled = [(m * Q + (t + 1) // 2) // t for m in pt]
        out = [[list(l) for l in p] for p in ct]
        for j, qj in enumerate(q):
            for k, s in enumerate(scaled):
                out[0][j][k] = (out[0][j][k] + s) % qj
        return out

    def mod_switch_to_next(self, ct, chain_idx):     # B8, big-int statement: floor((x + half)/q_last)
        q = self.base(chain_idx)
        ql, qn = q[-1], q[:-1]
        Qn = prod(qn)
        out = []
        for p in ct:
            x = self.crt(p, q)
            y = [((v + (ql >> 1)) // ql) % Qn for v in x]
            out.append([[v % qj for v in y] for qj in qn])
        return out

    def irrelevant_bit_count(self):
        return max(0, self.key_q[0].bit_length() - (self.t.bit_length() + self.n.bit_length() - 1))

    def clear_irrelevant_bits(self, ct):
        mask = ~((1 << self.irrelevant_bit_count()) - 1)
        return [[[v & mask for v in l] for l in p] for p in ct]

    # ---- BEHZ multiply (B9), every base conversion written as its big-integer definition
    @staticmethod
    def fastbconv_int(limbs_k, base):
        """sum_i [x_i (Q/q_i)^-1]_{q_i} (Q/q_i)  as an integer (= x + alpha*Q, 0 <= alpha < len)."""
        Q = prod(base)
        return sum(x * pow(Q // qi, -1, qi) % qi * (Q // qi) for x, qi in zip(limbs_k, base))

    def _extend(self, poly, chain_idx):
        """one poly [L][n] (coeff) -> (same poly in q, SmMRq'd poly in Bsk), both coefficient form."""
        q, Q, B, m_sk, _ = self.rns_tool(chain_idx)
        Bsk = B + [m_sk]
        mt = self.m_tilde
        out_bsk = [[0] * self.n for _ in Bsk]
        for k in range(self.n):
            xs = [poly[j][k] * mt % q[j] for j in range(len(q))]
            y = self.fastbconv_int(xs, q)
            r = (-(y % mt) * pow(Q, -1, mt)) % mt
            for i, m in enumerate(Bsk):
                rc = r + (m - mt) if r >= mt // 2 else r           # centred lift of r into Z_m
                out_bsk[i][k] = (y % m + Q * rc) * pow(mt, -1, m) % m
        return poly, out_bsk

    def multiply(self, a, b, chain_idx):
        q, Q, B, m_sk, _ = self.rns_tool(chain_idx)
        Bsk = B + [m_sk]
        t = self.t
        ea = [self._extend(p, chain_idx) for p in a]
        eb = [self._extend(p, chain_idx) for p in b]

        size = len(a) + len(b) - 1                   # no relinearisation between products: sizes add up (16 at most in SEAL)
        if size > 16:
            raise ValueError("invalid size")

        def tensor(which, base):
            A = [e[which] for e in ea]
            Bb = [e[which] for e in eb]
            d = [[] for _ in range(size)]
            for j, m in enumerate(base):
                for I in range(size):
                    acc = [0] * self.n
                    for i in range(max(0, I - (len(b) - 1)), min(I, len(a) - 1) + 1):
                        term = self.negacyclic_mul(A[i][j], Bb[I - i][j], m)
                        acc = [(u + v) % m for u, v in zip(acc, term)]
                    d[I].append(acc)
            return d

        dq, db = tensor(0, q), tensor(1, Bsk)
        Bprod = prod(B)
        out = []
        for p in range(size):
            res = [[0] * self.n for _ in q]
            for k in range(self.n):
                xq = [dq[p][j][k] * t % q[j] for j in range(len(q))]
                xb = [db[p][i][k] * t % Bsk[i] for i in range(len(Bsk))]
                # fast_floor: (x_Bsk - FastBConv(x_q)) * Q^-1 mod Bsk
                y = self.fastbconv_int(xq, q)
                fl = [(xb[i] - y) * pow(Q, -1, m) % m for i, m in enumerate(Bsk)]
                # fastbconv_sk
                z = self.fastbconv_int(fl[:-1], B)
                alpha = (z - fl[-1]) * pow(Bprod, -1, m_sk) % m_sk
                if alpha > m_sk // 2:
                    alpha -= m_sk
                for j, qj in enumerate(q):
                    res[j][k] = (z - alpha * Bprod) % qj
            out.append(res)
        return out

    def square(self, a, chain_idx):
        return self.multiply(a, a, chain_idx)

    # ---- relinearise (B10).  rk[i][comp][limb] NTT-form lists
    def relinearize(self, ct3, rk, chain_idx):
        q = self.base(chain_idx)
        L, K = len(q), self.K
        p = self.key_q[K - 1]
        c2 = ct3[2]
        moduli = q + [p]
        key_idx = list(range(L)) + [K - 1]
        acc = [[None] * (L + 1) for _ in range(2)]
        for I, (m, ki) in enumerate(zip(moduli, key_idx)):
            for comp in range(2):
                s = [0] * self.n
                for J in range(L):
                    tn = self.ntt([v % m for v in c2[J]], m)
                    key = rk[J][comp][ki]
                    s = [(u + x * y) % m for u, x, y in zip(s, tn, key)]
                acc[comp][I] = self.intt(s, m)
        out = []
        for comp in range(2):
            last = [(v + (p >> 1)) % p for v in acc[comp][L]]
            limbs = []
            for j, qj in enumerate(q):
                pinv = pow(p, -1, qj)
                limbs.append([
                    (ct3[comp][j][k] + (acc[comp][j][k] - last[k] + (p >> 1)) * pinv) % qj
                    for k in range(self.n)])
            out.append(limbs)
        return out

    # ------------------------------------------------------------------------- harness (B4 etc.)
    def slot_map(self):
        n, m = self.n, 2 * self.n
        row = n // 2
        mp, pos = [0] * n, 1
        for i in range(row):
            mp[i] = brv((pos - 1) >> 1, self.logn)
            mp[row | i] = brv((m - pos - 1) >> 1, self.logn)
            pos = pos * 3 % m
        return mp

    def encode(self, values):
        mp = self.slot_map()
        tmp = [0] * self.n
        for i, v in enumerate(values):
            tmp[mp[i]] = v
        return self.intt(tmp, self.t)

    def decode(self, pt):
        mp = self.slot_map()
        tmp = self.ntt(list(pt), self.t)
        return [tmp[mp[i]] for i in range(self.n)]

    def decrypt(self, s_coeff, ct, chain_idx):
        """s_coeff: ternary secret as ints in {-1,0,1}.  Returns (plaintext mod t, noise budget bits)."""
        q = self.base(chain_idx)
        Q, t = prod(q), self.t
        limbs = []
        for j, qj in enumerate(q):
            s = [v % qj for v in s_coeff]
            acc = [0] * self.n
            for p in reversed(ct[1:]):
                acc = self.negacyclic_mul([(x + y) % qj for x, y in zip(acc, p[j])], s, qj)
            limbs.append([(x + y) % qj for x, y in zip(acc, ct[0][j])])
        x = self.crt(limbs, q)
        pt, worst = [], 0
        for v in x:
            num = v * t
            m = (num + Q // 2) // Q
            rem = num % Q
            worst = max(worst, min(rem, Q - rem))
            pt.append(m % t)
        budget = Q.bit_length() if worst == 0 else max(0, (Q // (2 * worst)).bit_length() - 1)
        return pt, budget


# ----------------------------------------------------------------------------- path drivers
def create_powers_set(ps_low_degree, target_degree):
    if ps_low_degree:
        h = ps_low_degree + 1
        return list(range(1, ps_low_degree + 1)) + list(range(h, target_degree // h * h + 1, h))
    return list(range(1, target_degree + 1))


def powers_dag(sources, targets):
    sources, targets = sorted(sources), sorted(targets)
    tset = set(targets)
    depth, nodes = {}, []
    for cp in targets:
        if cp in sources:
            depth[cp] = 0
            nodes.append((cp, 0, 0, 0))
            continue
        best = (cp - 1, cp - 1, 1)
        for s1 in targets:
            if s1 >= cp:
                break
            s2 = cp - s1
            if s2 not in tset:
                continue
            d = max(depth[s1], depth[s2]) + 1
            if d < best[0]:
                best = (d, s1, s2)
        depth[cp] = best[0]
        nodes.append((cp, best[0], best[1], best[2]))
    return max(depth.values()), nodes


def coeff_is_ntt(ps_low_degree, i):
    return (not ps_low_degree and i != 0) or (bool(ps_low_degree) and i % (ps_low_degree + 1) != 0)


def compute_powers(M, sources, nodes, rk, ps_low_degree):
    """receiver_osn.cpp:395-488 on the model."""
    first = M.first
    pw = dict(sources)
    for d in range(1, max(nd[1] for nd in nodes) + 1):
        for power, depth, p1, p2 in nodes:
            if depth != d:
                continue
            prod3 = M.multiply(pw[p1], pw[p2], first)
            pw[power] = M.relinearize(prod3, rk, first) if M.K > 1 else prod3
    high, low = M.clamp(1), M.clamp(2)
    out = {}
    for power, _, _, _ in nodes:
        ct, lvl = pw[power], first
        target = high if (not ps_low_degree or power > ps_low_degree) else low
        while lvl > target:
            ct = M.mod_switch_to_next(ct, lvl)
            lvl -= 1
        if not ps_low_degree or power <= ps_low_degree:
            ct = M.transform_to_ntt(ct, lvl)
        out[power] = ct
    return out


def eval_plain(M, powers, coeffs, lvl, mask):
    """bin_bundle.cpp:106-174."""
    result = [[[0] * M.n for _ in M.base(lvl)] for _ in range(2)]      # :132-134: size 2, zero
    for deg in range(1, len(coeffs)):
        result = M.add(result, M.multiply_plain_ntt(powers[deg], coeffs[deg], lvl), lvl)
    result = M.transform_from_ntt(result, lvl)
    result = M.add_plain(result, coeffs[0], lvl)
    result = M.add_plain(result, mask, lvl)
    while lvl > 0:
        result = M.mod_switch_to_next(result, lvl)
        lvl -= 1
    return M.clear_irrelevant_bits(result)


def eval_patstock(M, powers, coeffs, ps_low_degree, rk, mask):
    """bin_bundle.cpp:192-360."""
    degree = len(coeffs) - 1
    assert 1 < ps_low_degree < degree
    high = M.clamp(1)
    low = min(M.first, 2)
    h = ps_low_degree + 1
    H = degree // h
    zero3 = [[[0] * M.n for _ in M.base(high)] for _ in range(3)]
    result = zero3
    for i in range(1, H + 1):
        jmax = h - 1 if i < H else degree % h
        if jmax == 0:
            break
        inner = None
        for j in range(1, jmax + 1):
            term = M.multiply_plain_ntt(powers[j], coeffs[i * h + j], low)
            inner = term if inner is None else M.add(inner, term, low)
        inner = M.transform_from_ntt(inner, low)
        for l in range(low, high, -1):
            inner = M.mod_switch_to_next(inner, l)
        result = M.add(result, M.multiply(inner, powers[i * h], high), high)
    if M.K > 1:
        result = M.relinearize(result, rk, high)
    for j in range(1, h):
        term = M.multiply_plain_ntt(powers[j], coeffs[j], low)
        term = M.transform_from_ntt(term, low)
        for l in range(low, high, -1):
            term = M.mod_switch_to_next(term, l)
        result = M.add(result, term, high)
    for i in range(1, H + 1):
        result = M.add(result, M.multiply_plain_coeff(powers[i * h], coeffs[i * h], high), high)
    result = M.add_plain(result, coeffs[0], high)
    result = M.add_plain(result, mask, high)
    for l in range(high, 0, -1):
        result = M.mod_switch_to_next(result, l)
    return M.clear_irrelevant_bits(result)


def vec_to_oc_block(values, felts_per_item, plain_modulus):
    """receiver/apsu/receiver_osn.cpp:53-73 with Python integers; uint64_t wrap-around made explicit.
    -> (lower, higher), the two 64-bit halves handed to oc::toBlock(higher, lower)."""
    M = (1 << 64) - 1
    ln = 1
    while ((1 << ln) - 1) < plain_modulus:
        ln += 1
    mask = (1 << ln) - 1
    mask_lower = (1 << (ln >> 1)) - 1
    mask_higher = mask - mask_lower
    lower = higher = 0
    if felts_per_item & 1:
        lower = values[felts_per_item - 1] & mask_lower
        higher = (values[felts_per_item - 1] & mask_higher) >> ((ln >> 1) - 1)
    pla = 0
    while pla < felts_per_item - 1:
        lower = ((values[pla] & mask) | (lower << ln)) & M
        higher = ((values[pla + 1] & mask) | (higher << ln)) & M
        pla += 2
    return lower, higher
