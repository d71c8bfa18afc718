"""A second, independent restatement of the reference's two-layer trainer -- pure Python, small
inputs only -- written straight from the reference source to pin the C oracle (oracle/) from
another side.  TEST INFRASTRUCTURE: imported by tests/test_pyref.py only.

It follows the reference literally (vectors of (key, y) pairs, iterators, Option as None) rather
than the closed forms the oracle and the kernels use, so a slip in one is unlikely to be repeated in
the other.  u64, u32 and f64 keys.  Python floats are IEEE doubles and nothing is contracted; `mul_add` is
computed exactly in rationals and rounded once; `powf(3.0)` is libm's pow like the reference's,
`powf(2.0)` is x*x (what LLVM makes of it).  Where the reference asserts or unwraps None this code
raises ReferencePanic.

Citations are file:line in the reference repository (rmi_lib/src/...).
"""
import math
from fractions import Fraction

U64 = (1 << 64) - 1


class ReferencePanic(Exception):
    pass


def _assert(cond, what=""):
    if not cond:
        raise ReferencePanic(what)


def fma(a, b, c):                                   # f64::mul_add: one rounding
    if any(math.isinf(v) or math.isnan(v) for v in (a, b, c)):
        return a * b + c
    exact = Fraction(a) * Fraction(b) + Fraction(c)
    try:
        return float(exact)
    except OverflowError:                           # beyond the largest double: rounds to infinity
        return math.inf if exact > 0 else -math.inf


def sat_u64(f):                                     # Rust `f64 as u64`
    if f != f or f <= 0.0:
        return 0
    if f >= 18446744073709551616.0:
        return U64
    return int(f)


# ---- TrainingKey (models/mod.rs:65-111) for the three key types of src/load.rs ----
class KeyType:
    def __init__(self, kind):
        self.kind = kind
        self.max_value = {"u64": U64, "u32": (1 << 32) - 1, "f64": 1.7976931348623157e308}[kind]
        self.zero_value = 0.0 if kind == "f64" else 0

    def minus_epsilon(self, k):                     # unchecked -1 (release build: wraps)
        if self.kind == "f64":
            return k - 2.220446049250313e-16
        return (k - 1) & (U64 if self.kind == "u64" else (1 << 32) - 1)

    def plus_epsilon(self, k):
        if self.kind == "f64":
            return k + 2.220446049250313e-16
        return (k + 1) & (U64 if self.kind == "u64" else (1 << 32) - 1)


def as_int(key):                                    # ModelInput::as_int, models/mod.rs:428-433
    return sat_u64(key) if isinstance(key, float) else key


# ---- models/mod.rs: RMITrainingData over a vector of (key, offset) pairs ----
class Data:
    def __init__(self, pairs, scale=1.0):
        self.pairs = list(pairs)
        self.scale = scale

    def __len__(self):
        return len(self.pairs)

    def _map(self, key, off):                       # map_scale!, models/mod.rs:238-250
        if abs(self.scale - 1.0) > 2.220446049250313e-16:
            return key, int(float(off) * self.scale)
        return key, off

    def get(self, idx):                             # :268-270
        return self._map(*self.pairs[idx])

    def get_key(self, idx):                         # :272-274
        return self.pairs[idx][0]

    def iter(self):                                 # :276-278 over FixDupsIter :154-185
        last = None
        it = iter(self.pairs)
        while True:
            if last is None:
                nxt = next(it, None)
                if nxt is None:
                    return
                last = nxt
                yield self._map(*nxt)
            else:
                nxt = next(it, None)
                if nxt is not None:
                    if nxt[0] == last[0]:
                        yield self._map(nxt[0], last[1])
                    else:
                        last = nxt
                        yield self._map(*nxt)
                else:
                    item, last = last, None         # self.last_item.take(): the last item once more
                    yield self._map(*item)
                    # (the iterator is exhausted: the next call finds last_item None and the inner iterator empty)
                    return

    def lower_bound_by(self, f):                    # :294-309
        size = len(self)
        if size == 0:
            return 0
        base = 0
        while size > 1:
            half = size // 2
            mid = base + half
            if f(self.get(mid)) < 0:
                base = mid
            size -= half
        return base + (1 if f(self.get(base)) < 0 else 0)


# ---- models/linear.rs ----
def slr(items):                                     # :12-59
    mean_x = mean_y = c = m2 = 0.0
    n = 0
    for x, y in items:
        n += 1
        dx = x - mean_x
        mean_x += dx / float(n)
        mean_y += (y - mean_y) / float(n)
        c += dx * (y - mean_y)
        dx2 = x - mean_x
        m2 += dx * dx2
    if n == 0:
        return 0.0, 0.0
    if n == 1:
        return mean_y, 0.0
    cov = c / float(n - 1)
    var = m2 / float(n - 1)
    _assert(var >= 0.0, "negative variance")
    if var == 0.0:
        return mean_y, 0.0
    beta = cov / var
    return mean_y - beta * mean_x, beta


class Linear:                                       # :75-120
    def __init__(self, data):
        self.p = slr((float(x), float(y)) for x, y in data.iter())

    def predict_to_float(self, key):
        return fma(self.p[1], float(key), self.p[0])

    needs_bounds_check = True

    def set_to_constant_model(self, c):
        self.p = (float(c), 0.0)
        return True

    def params(self):
        return list(self.p)


class RobustLinear(Linear):                         # :233-261
    def __init__(self, data):
        total = len(data)
        if total == 0:
            self.p = (0.0, 0.0)
            return
        bnd = max(1, int(float(total) * 0.0001))
        _assert(bnd * 2 + 1 < total, "robust_linear needs more data")
        items = list(data.iter())[bnd:bnd + (total - 2 * bnd)]
        self.p = slr((float(x), float(y)) for x, y in items)


# ---- models/linear_spline.rs ----
def linear_splines(data):                           # :13-35
    if len(data) == 0:
        return 0.0, 0.0
    if len(data) == 1:
        return float(data.get(0)[1]), 0.0
    first, last = data.get(0), data.get(len(data) - 1)
    if first[0] == last[0]:
        return float(data.get(0)[1]), 0.0
    slope = (float(first[1]) - float(last[1])) / (float(first[0]) - float(last[0]))
    intercept = float(first[1]) - slope * float(first[0])
    return intercept, slope


class LinearSpline(Linear):
    def __init__(self, data):
        self.p = linear_splines(data)


# ---- models/cubic_spline.rs ----
def _scale(v, lo, hi):                              # scale!, :11-15
    return (v - lo) / (hi - lo)


def cubic(data):                                    # :18-101
    if len(data) == 0:
        return 0.0, 0.0, 1.0, 0.0
    if len(data) == 1:
        return 0.0, 0.0, 0.0, float(data.get(0)[1])
    candidate = data.get(0)[0]
    if not any(x != candidate for x, _ in data.iter()):
        return 0.0, 0.0, 0.0, float(data.get(0)[1])
    first, last = data.get(0), data.get(len(data) - 1)
    xmin, ymin = float(first[0]), float(first[1])
    xmax, ymax = float(last[0]), float(last[1])
    nxt = next(((tx, ty) for tx, ty in data.iter() if _scale(float(tx), xmin, xmax) > 0.0), None)
    _assert(nxt is not None, "cubic: unwrap on None (:50)")
    sxn, syn = _scale(float(nxt[0]), xmin, xmax), _scale(float(nxt[1]), ymin, ymax)
    m1 = (syn - 0.0) / (sxn - 0.0)
    prv = next((data.get(i) for i in range(len(data) - 1, -1, -1) if _scale(float(data.get(i)[0]), xmin, xmax) < 1.0), None)
    _assert(prv is not None, "cubic: unwrap on None (:61)")
    sxp, syp = _scale(float(prv[0]), xmin, xmax), _scale(float(prv[1]), ymin, ymax)
    m2 = (1.0 - syp) / (1.0 - sxp)
    if m1 * m1 + m2 * m2 > 9.0:
        tau = 3.0 / math.sqrt(m1 * m1 + m2 * m2)
        m1 *= tau
        m2 *= tau
    cube = math.pow(xmax - xmin, 3.0)
    a = (m1 + m2 - 2.0) / cube
    b = -(xmax * (2.0 * m1 + m2 - 3.0) + xmin * (m1 + 2.0 * m2 - 3.0)) / cube
    c = (m1 * (xmax * xmax) + m2 * (xmin * xmin) + xmax * xmin * (2.0 * m1 + 2.0 * m2 - 6.0)) / cube
    d = -xmin * (m1 * (xmax * xmax) + xmax * xmin * (m2 - 3.0) + xmin * xmin) / cube
    a *= ymax - ymin
    b *= ymax - ymin
    c *= ymax - ymin
    d *= ymax - ymin
    d += ymin
    return a, b, c, d


class Cubic:                                        # :103-192
    needs_bounds_check = False

    def __init__(self, data):
        self.p = cubic(data)
        lin = LinearSpline(data)
        our = lin_err = 0.0
        for x, y in data.iter():                    # iter_model_input
            our += abs(self.predict_to_float(x) - float(y))
            lin_err += abs(lin.predict_to_float(x) - float(y))
        if lin_err < our:
            self.p = (0.0, 0.0, lin.p[1], lin.p[0])

    def predict_to_float(self, key):
        a, b, c, d = self.p
        v = float(key)
        return fma(fma(fma(a, v, b), v, c), v, d)

    def set_to_constant_model(self, c):
        self.p = (0.0, 0.0, 0.0, float(c))
        return True

    def params(self):
        return list(self.p)


# ---- models/radix.rs, utils.rs ----
def num_bits(largest):                              # utils.rs:13-21
    nbits = 0
    while (1 << (nbits + 1)) - 1 <= largest:
        nbits += 1
    _assert(nbits >= 1, "num_bits")
    return nbits


def common_prefix_size(data):                       # utils.rs:23-36
    any_ones, no_ones = 0, U64
    for x, _ in data.iter():
        any_ones |= as_int(x)
        no_ones &= as_int(x)
    any_zeros = ~no_ones & U64
    prefix_bits = any_zeros ^ any_ones
    inv = ~prefix_bits & U64
    return 64 - inv.bit_length()                    # leading_zeros


class Radix:                                        # radix.rs:13-81
    needs_bounds_check = False

    def __init__(self, data):
        if len(data) == 0:
            self.ip = (0, 0)
            return
        bits = num_bits(max(y for _, y in data.iter()))
        self.ip = (common_prefix_size(data), bits)

    def predict_to_int(self, key):
        prefix, bits = self.ip
        return ((as_int(key) << (prefix & 63)) & U64) >> ((64 - bits) & 63)


def predict_to_int(model, key):                     # models/mod.rs:735-737: f64::max(0.0, pred.floor()) as u64
    if isinstance(model, Radix):
        return model.predict_to_int(key)
    f = model.predict_to_float(key)
    if f != f:
        return 0                                    # f64::max(0.0, NaN) == 0.0
    if abs(f) < 4503599627370496.0:                 # below 2^52 floor() can change the value
        f = float(math.floor(f))
    return sat_u64(max(0.0, f))


# ---- models/normal.rs, LogLinearModel in models/linear.rs (used as roots) ----
def exp1(x):                                        # normal.rs:12-23, linear.rs:156-166
    x = 1.0 + x / 64.0
    for _ in range(6):
        x *= x
    return x


def phi(x):                                         # normal.rs:25-27
    return 1.0 / (1.0 + exp1(-1.65451 * x))


class Normal:                                       # normal.rs:29-50, :70-126
    needs_bounds_check = True

    def __init__(self, data):
        scale, mean, stdev = -math.inf, 0.0, 0.0
        n = float(len(data))
        for x, y in data.iter():
            mean += float(x) / n
            scale = max(scale, float(y))
        for x, _ in data.iter():
            stdev += (float(x) - mean) * (float(x) - mean)
        stdev /= n
        self.p = (mean, math.sqrt(stdev), scale)

    def predict_to_float(self, key):
        mean, stdev, scale = self.p
        try:
            return phi((float(key) - mean) / stdev) * scale
        except ZeroDivisionError:
            return math.nan

    def params(self):
        return list(self.p)


class LogLinear(Linear):                            # linear.rs:60-72, :152-210
    def __init__(self, data):
        pts = [(float(x), math.log(float(y))) for x, y in data.iter() if y > 0]     # ln 0 = -inf is filtered out
        self.p = slr(pts)

    def predict_to_float(self, key):
        return exp1(fma(self.p[1], float(key), self.p[0]))


MODELS = {"linear": Linear, "robust_linear": RobustLinear, "linear_spline": LinearSpline, "cubic": Cubic, "radix": Radix,
          "normal": Normal, "loglinear": LogLinear}


def train_model(name, data):                        # train/mod.rs:35-57
    return MODELS[name](data)


# ---- train/lower_bound_correction.rs ----
class LowerBoundCorrection:                         # :92-137
    def __init__(self, pred, num_leaves, data, kt):
        L = num_leaves
        first, last, runs = [None] * L, [None] * L, [0] * L
        last_target, run_len, run_key = 0, 0, data.get_key(0)
        for x, y in data.iter():
            target = min(L - 1, pred(x))
            if target == last_target and x == run_key:
                run_len += 1
            elif target != last_target or x != run_key:
                runs[last_target] = max(runs[last_target], run_len)
                run_len, run_key, last_target = 1, x, target
            if first[target] is None:
                first[target] = (y, x)
            last[target] = (y, x)
        n = len(data)
        nxt = [(0, kt.zero_value)] * L              # compute_next_for_leaf :30-56
        idx = 0
        while idx < L:
            above = None
            if idx != L - 1:
                for i in range(idx + 1, L):
                    if first[i] is not None:
                        above = (i, first[i])
                        break
            if above is not None:
                for i in range(idx, above[0]):
                    nxt[i] = above[1]
                idx = above[0]
            else:
                for i in range(idx, L):
                    nxt[i] = (n, kt.max_value)
                break
        prv = [(0, kt.zero_value)] * L              # compute_prev_for_leaf :58-80
        idx = L - 1
        while idx > 0:
            below = None
            for i in range(idx - 1, -1, -1):
                if last[i] is not None:
                    below = (i, last[i])
                    break
            if below is None:
                break
            for i in range(below[0] + 1, idx + 1):
                prv[i] = below[1]
            idx = below[0]
        self.first, self.last, self.next, self.prev, self.runs = first, last, nxt, prv, runs


# ---- train/two_layer.rs ----
def error_between(v1, v2, max_pred):                # :14-18
    p1, p2 = min(v1, max_pred), min(v2, max_pred)
    return max(p1, p2) - min(p1, p2)


def build_models_from(data, top, model_type, start_idx, end_idx, first_model_idx, num_models):   # :20-99
    _assert(end_idx > start_idx, "degenerate split")
    _assert(end_idx <= len(data) and start_idx <= len(data))
    leaf_models, second, last_target = [], [], first_model_idx
    items = list(data.iter())[start_idx:start_idx + (end_idx - start_idx)]
    for x, y in items:
        pred = predict_to_int(top, x)
        _assert(top.needs_bounds_check or pred < first_model_idx + num_models, "root out of bounds")
        target = min(first_model_idx + num_models - 1, pred)
        _assert(target >= last_target, "non-monotone")
        if target > last_target:
            last_item = second[-1] if second else None
            second.append((x, y))
            leaf_models.append(train_model(model_type, Data(second)))
            for _ in range(last_target + 1, target):
                leaf_models.append(train_model(model_type, Data([])))
            _assert(len(leaf_models) + first_model_idx == target)
            second = []
            if last_item is not None:
                second.append(last_item)
        second.append((x, y))
        last_target = target
    _assert(len(second) > 0)
    leaf_models.append(train_model(model_type, Data(second)))
    _assert(len(leaf_models) <= num_models)
    for _ in range(last_target + 1, first_model_idx + num_models):
        leaf_models.append(train_model(model_type, Data([])))
    _assert(len(leaf_models) == num_models)
    return leaf_models


def train_two_layer(keys, layer1, layer2, num_leaves, kind="u64"):   # :101-306
    L = num_leaves
    kt = KeyType(kind)
    md = Data([((float(k) if kind == "f64" else int(k)), i) for i, k in enumerate(keys)])
    n = len(md)
    md.scale = float(L) / float(n)
    top = train_model(layer1, md)
    md.scale = 1.0
    mid = L // 2

    def cmp_mid(item):
        t = min(L - 1, predict_to_int(top, item[0]))
        return -1 if t < mid else (0 if t == mid else 1)
    split_idx = md.lower_bound_by(cmp_mid)
    if 0 < split_idx < n:
        _assert(predict_to_int(top, md.get_key(split_idx)) > predict_to_int(top, md.get_key(split_idx - 1)))
    if split_idx >= n:
        leaves = build_models_from(md, top, layer2, 0, n, 0, L)
    else:
        st = min(L - 1, predict_to_int(top, md.get_key(split_idx)))
        leaves = build_models_from(md, top, layer2, 0, split_idx, 0, st) + \
            build_models_from(md, top, layer2, split_idx + 1, n, st, L - st)
    lb = LowerBoundCorrection(lambda x: predict_to_int(top, x), L, md, kt)
    for idx in range(L - 1):
        _assert((lb.first[idx] is None) == (lb.last[idx] is None))
        if lb.last[idx] is None:
            leaves[idx].set_to_constant_model(lb.next[idx][0])
    l1 = [(0, 0)] * L
    for x, y in md.iter():
        target = min(L - 1, predict_to_int(top, x))
        err = error_between(predict_to_int(leaves[target], x), y, n)
        l1[target] = (l1[target][0] + 1, max(err, l1[target][1]))
    for leaf in range(L):
        curr = l1[leaf][1]
        idx_next, key_next = lb.next[leaf]
        upper = error_between(predict_to_int(leaves[leaf], kt.minus_epsilon(key_next)), idx_next + 1, n)
        prev_idx = 0 if leaf == 0 else leaf - 1
        first_idx = lb.next[prev_idx][0]
        lower = error_between(predict_to_int(leaves[leaf], kt.plus_epsilon(lb.prev[leaf][1])), first_idx, n)
        l1[leaf] = (l1[leaf][0], max(curr, upper, lower) + lb.runs[leaf])
    m_idx, m_err = 0, l1[0][1]
    for i, (_, e) in enumerate(l1):                 # max_by_key: the last maximum
        if e >= m_err:
            m_idx, m_err = i, e
    avg = float(sum(c * e for c, e in l1) & U64) / float(n)
    return {"root": top, "leaves": leaves, "counts": [c for c, _ in l1], "errs": [e for _, e in l1],
            "max_error": m_err, "max_error_idx": m_idx, "avg_error": avg}
