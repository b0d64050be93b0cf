"""Exact symbolic derivatives of thermodynamic averages as Laurent polynomials in
moment "atoms", compiled to the table format libtxmom evaluates on the device.

The reference builds these with sympy ``Function`` subclasses whose ``fdiff``
encodes d/d(beta) of each average, expands the result and ``lambdify``-s it
(/root/reference/src/thermoextrap/beta.py:32-266, models.py:103-257).  Here the
same recursion rules act on a tiny polynomial algebra:

    atom      a hashable tuple, e.g. ("du", 3), ("dxdu", 2, 1), ("x1", None)
    monomial  sorted tuple of (atom, integer power), power may be negative
    Poly      {monomial: Fraction}  (+ an optional "-log(atom)" term)

``d/dbeta`` is the product/chain rule over atoms with a per-family rule table, so
new families (lnPi, chained -log) are one small function each.  Coefficients
stay exact integers/rationals until the table is written.
"""

from __future__ import annotations

from fractions import Fraction
from typing import Callable, Iterable

Atom = tuple
Mono = tuple  # tuple[(Atom, int), ...] sorted


def _key(atom):
    # None sorts before ints inside atoms
    return tuple((0, "") if v is None else ((1, v) if isinstance(v, int) else (2, str(v))) for v in atom)


def _mono_mul(a: Mono, b: Mono) -> Mono:
    d = dict(a)
    for at, p in b:
        q = d.get(at, 0) + p
        if q:
            d[at] = q
        else:
            d.pop(at, None)
    return tuple(sorted(d.items(), key=lambda kv: _key(kv[0])))


class Poly:
    """Laurent polynomial in atoms, optionally plus ``-log(log_atom)``."""

    __slots__ = ("terms", "log_atom")

    def __init__(self, terms: dict | None = None, log_atom: Atom | None = None):
        self.terms: dict[Mono, Fraction] = {}
        for m, c in (terms or {}).items():
            if c:
                self.terms[m] = Fraction(c)
        self.log_atom = log_atom

    # constructors
    @classmethod
    def const(cls, c) -> "Poly":
        return cls({(): Fraction(c)})

    @classmethod
    def atom(cls, a: Atom, power: int = 1) -> "Poly":
        return cls({((a, power),): Fraction(1)})

    @classmethod
    def minus_log(cls, a: Atom) -> "Poly":
        return cls({}, log_atom=a)

    # algebra
    def __add__(self, o):
        o = _as_poly(o)
        if self.log_atom is not None and o.log_atom is not None:
            raise NotImplementedError("sum of two log terms")
        t = dict(self.terms)
        for m, c in o.terms.items():
            v = t.get(m, 0) + c
            if v:
                t[m] = v
            else:
                t.pop(m, None)
        return Poly(t, self.log_atom if self.log_atom is not None else o.log_atom)

    __radd__ = __add__

    def __neg__(self):
        if self.log_atom is not None:
            raise NotImplementedError("negating a log term")
        return Poly({m: -c for m, c in self.terms.items()})

    def __sub__(self, o):
        return self + (-_as_poly(o))

    def __mul__(self, o):
        o = _as_poly(o)
        if self.log_atom is not None or o.log_atom is not None:
            raise NotImplementedError("product with a log term")
        t: dict[Mono, Fraction] = {}
        for m1, c1 in self.terms.items():
            for m2, c2 in o.terms.items():
                m = _mono_mul(m1, m2)
                v = t.get(m, 0) + c1 * c2
                if v:
                    t[m] = v
                else:
                    t.pop(m, None)
        return Poly(t)

    __rmul__ = __mul__

    def __pow__(self, k: int):
        if k < 0:
            if len(self.terms) != 1 or self.log_atom is not None:
                raise NotImplementedError("negative power of a sum")
            ((m, c),) = self.terms.items()
            return Poly({tuple((a, p * k) for a, p in m): Fraction(c) ** k})
        out = Poly.const(1)
        for _ in range(k):
            out = out * self
        return out

    def __eq__(self, o):
        o = _as_poly(o)
        return self.terms == o.terms and self.log_atom == o.log_atom

    def __hash__(self):
        return hash((frozenset(self.terms.items()), self.log_atom))

    def atoms(self) -> list[Atom]:
        s = {a for m in self.terms for a, _ in m}
        if self.log_atom is not None:
            s.add(self.log_atom)
        return sorted(s, key=_key)

    def is_zero(self) -> bool:
        return not self.terms and self.log_atom is None

    def diff(self, rule: Callable[[Atom], "Poly"]) -> "Poly":
        """d/d(alpha): sum over terms and factors of  c * p * a^(p-1) * rule(a) * rest."""
        out = Poly()
        cache: dict[Atom, Poly] = {}

        def da(a):
            if a not in cache:
                cache[a] = rule(a)
            return cache[a]

        for m, c in self.terms.items():
            for i, (a, p) in enumerate(m):
                rest = m[:i] + (((a, p - 1),) if p != 1 else ()) + m[i + 1:]
                out = out + Poly({rest: c * p}) * da(a)
        if self.log_atom is not None:
            out = out + Poly({((self.log_atom, -1),): Fraction(-1)}) * da(self.log_atom)
        return out

    def subs(self, mapping: Callable[[Atom], "Poly | None"]) -> "Poly":
        """Replace atoms by polynomials (mapping returns None to keep an atom)."""
        out = Poly()
        for m, c in self.terms.items():
            term = Poly.const(c)
            for a, p in m:
                r = mapping(a)
                term = term * ((Poly.atom(a) if r is None else r) ** p)
            out = out + term
        if self.log_atom is not None:
            r = mapping(self.log_atom)
            if r is not None:
                raise NotImplementedError("substitution inside log")
            out = out + Poly.minus_log(self.log_atom)
        return out

    def __repr__(self):
        parts = []
        for m, c in sorted(self.terms.items(), key=lambda kv: [(_key(a), p) for a, p in kv[0]]):
            f = "*".join(f"{_atom_str(a)}" + (f"**{p}" if p != 1 else "") for a, p in m) or "1"
            parts.append(f"{c}*{f}" if c != 1 or not m else f)
        if self.log_atom is not None:
            parts.append(f"-log({_atom_str(self.log_atom)})")
        return " + ".join(parts) if parts else "0"


def _as_poly(o) -> Poly:
    return o if isinstance(o, Poly) else Poly.const(o)


def _atom_str(a: Atom) -> str:
    idx = [str(v) for v in a[1:] if v is not None]
    return a[0] + (f"[{','.join(idx)}]" if idx else "")


# ---------------------------------------------------------------------------
# atom families and their beta-derivative rules
# (reference beta.py:52-54, 83-85, 110-116, 144-151, 174-176, 193-196, 216-218, 246-258)
# ---------------------------------------------------------------------------
def du(n: int) -> Poly:
    """<(u - <u>)^n>  with du(0) = 1, du(1) = 0."""
    if n == 0:
        return Poly.const(1)
    if n == 1:
        return Poly.const(0)
    return Poly.atom(("du", n))


def dxdu(n: int, d=None) -> Poly:
    """<dx^(d) du^n>  with dxdu(0, .) = 0."""
    if n == 0:
        return Poly.const(0)
    return Poly.atom(("dxdu", n, d))


def x1(d=None) -> Poly:
    return Poly.atom(("x1", d))


def umean() -> Poly:
    return Poly.atom(("umean",))


def u_raw(n: int) -> Poly:
    """<u^n> with u(0) = 1."""
    return Poly.const(1) if n == 0 else Poly.atom(("u", n))


def xu_raw(n: int, d=None) -> Poly:
    return Poly.atom(("xu", n, d))


def beta_rule(atom: Atom) -> Poly:
    """d/d(beta) of one atom at fixed sample set (canonical ensemble weights e^{-beta u})."""
    kind = atom[0]
    if kind == "du":
        n = atom[1]
        return -du(n + 1) + n * du(n - 1) * du(2)
    if kind == "dxdu":
        _, n, d = atom
        out = -dxdu(n + 1, d) + n * dxdu(n - 1, d) * du(2) + dxdu(1, d) * du(n)
        if d is not None:
            out = out + dxdu(n, d + 1)
        return out
    if kind == "x1":
        d = atom[1]
        out = -dxdu(1, d)
        if d is not None:
            out = out + x1(d + 1)
        return out
    if kind == "umean":
        return -du(2)
    if kind == "u":
        n = atom[1]
        return -u_raw(n + 1) + u_raw(n) * u_raw(1)
    if kind == "xu":
        _, n, d = atom
        out = -xu_raw(n + 1, d) + xu_raw(n, d) * u_raw(1)
        if d is not None:
            out = out + xu_raw(n, d + 1)
        return out
    if kind == "const":  # quantities that do not depend on beta (volume, mu*N, ...)
        return Poly.const(0)
    raise ValueError(f"no beta-derivative rule for atom {atom}")


def chain_rule(atom: Atom) -> Poly:
    """X[k] -> X[k+1]: used to push a post-transform through given derivatives
    (the -log<X> chain rule of reference models.py:261-288)."""
    if atom[0] != "X":
        raise ValueError(atom)
    return Poly.atom(("X", atom[1] + 1))


class DerivSeries:
    """Lazy sequence  f, f', f'', ...  under a rule (reference models.SymDerivBase)."""

    def __init__(self, func: Poly, rule: Callable[[Atom], Poly] = beta_rule, post_func=None):
        self.func_orig = func
        self.rule = rule
        self.post_func = post_func
        self._items = [apply_post_func(func, post_func)]

    def __getitem__(self, order: int) -> Poly:
        while len(self._items) <= order:
            self._items.append(self._items[-1].diff(self.rule))
        return self._items[order]


def apply_post_func(func: Poly, post_func) -> Poly:
    """post_func in {None, 'minus_log', 'pow_k', callable(Poly) -> Poly}."""
    if post_func is None:
        return func
    if isinstance(post_func, str):
        if post_func == "minus_log":
            if len(func.terms) != 1 or func.log_atom is not None:
                raise NotImplementedError("minus_log of a non-atomic average")
            ((m, c),) = func.terms.items()
            if c != 1 or len(m) != 1 or m[0][1] != 1:
                raise NotImplementedError("minus_log of a non-atomic average")
            return Poly.minus_log(m[0][0])
        if post_func.startswith("pow_"):
            return func ** int(post_func.split("_")[-1])
        raise ValueError("post_func must be callable or in {minus_log, pow_1, pow_2, ...}")
    # A callable.  The reference hands it the sympy function and differentiates what comes back (models.py:124-137), so
    # callables written for it are sympy -> sympy (``lambda f: f**2``, ``lambda f: -sp.log(f)``): call it with a sympy symbol
    # first and translate the expression -- a Laurent polynomial in f with rational coefficients, optionally one -log(f)
    # term: what the table evaluator can represent.  A callable written for this package's Poly type gets the Poly.
    import sympy as sp

    fs = sp.Symbol("f", positive=True)
    try:
        expr = post_func(fs)
    except Exception:  # noqa: BLE001 -- not a sympy-style callable
        expr = None
    if isinstance(expr, sp.Basic):
        return poly_from_sympy(expr, fs, func)
    return post_func(func)


def poly_from_sympy(expr, fs, func: Poly) -> Poly:
    """``expr`` (sympy, in the symbol ``fs``) with ``func`` substituted for the symbol, as a Poly: sum_k c_k f^k with
    integer k (negative allowed) and rational c_k, plus at most one ``-log(f)`` (coefficient exactly -1: what the table's
    TXM_FUNC_MINUS_LOG flag evaluates).  Anything else (exp, sqrt, other functions of f) raises NotImplementedError."""
    import sympy as sp
    from fractions import Fraction

    expr = sp.expand(expr)
    out = None
    rest = sp.Integer(0)
    for term in sp.Add.make_args(expr):
        if term.has(sp.log):
            if sp.simplify(term + sp.log(fs)) != 0:
                raise NotImplementedError(f"post_func: only a bare -log(f) term is supported, got {term}")
            if out is not None:
                raise NotImplementedError("post_func: more than one log term")
            out = apply_post_func(func, "minus_log")
        else:
            rest += term
    poly = None
    for term in sp.Add.make_args(sp.expand(rest)):
        if term == 0:
            continue
        c, k = term.as_coeff_exponent(fs)
        if c.has(fs) or not (k.is_Integer and c.is_Rational):
            raise NotImplementedError(f"post_func: {term} is not a rational multiple of an integer power of f")
        cf = Fraction(int(c.p), int(c.q))
        k = int(k)
        t = (Poly.const(1) if k == 0 else func ** k) * cf
        poly = t if poly is None else poly + t
    if out is None:
        if poly is None:
            raise NotImplementedError("post_func returned 0")
        return poly
    return out if poly is None else out + poly


class NotRepresentable(NotImplementedError):
    """An expression outside the table evaluator's algebra (Laurent polynomials in atoms + one ``-log(atom)``), or a
    symbol with no device source: ``Derivatives`` then evaluates on the host."""


def _atom_from_sympy(obj) -> Atom | None:
    """A sympy ``Indexed`` / ``Symbol`` of the reference's naming (beta.py:46-240, volume.py:63-78) -> atom, None for a
    symbol occurrence that is identically 1 (``du[0]``, ``u[0]``); raises for ``du[1]`` / ``dxdu[0]`` = 0 via ZeroAtom."""
    import sympy as sp

    if isinstance(obj, sp.Indexed):
        name = _ROLE.get(str(obj.base.label), str(obj.base.label))
        try:
            idx = tuple(int(i) for i in obj.indices)
        except TypeError as e:
            raise NotRepresentable(f"symbolic index in {obj}") from e
        if name in ("du", "u"):
            if len(idx) != 1:
                raise NotRepresentable(f"{obj}: one index expected")
            return (name, idx[0])
        if name in ("dxdu", "xu"):
            if len(idx) not in (1, 2):
                raise NotRepresentable(f"{obj}: one or two indices expected")
            return (name, idx[0], idx[1] if len(idx) == 2 else None)
        if name == "x1":
            return ("x1", idx[0])
        return (name, *idx)
    if isinstance(obj, sp.Symbol):
        name = str(obj.name)
        if name == "u":
            return ("umean",)
        if name == "x1":
            return ("x1", None)
        return (name,)
    raise NotRepresentable(f"{obj} is not a symbol of the moment families")


def _atom_value(a: Atom):
    """identities of the families: du[0] = 1, du[1] = 0, dxdu[0] = 0, u[0] = 1 (reference beta.py:57-66, 119-126)."""
    if a[0] == "du" and a[1] in (0, 1):
        return 1 - a[1]
    if a[0] == "u" and a[1] == 0:
        return 1
    if a[0] == "dxdu" and a[1] == 0:
        return 0
    return None


def poly_from_expr(expr) -> Poly:
    """A sympy expression over the reference's symbols as a Poly; NotRepresentable when it is not a Laurent polynomial
    with rational coefficients plus at most one bare ``-log(symbol)``."""
    import sympy as sp

    expr = sp.expand(sp.sympify(expr))
    out = Poly()
    for term in sp.Add.make_args(expr):
        c, rest = term.as_coeff_Mul()
        if not c.is_Rational:
            raise NotRepresentable(f"coefficient {c} is not rational")
        if isinstance(rest, sp.log):
            if c != -1:
                raise NotRepresentable(f"only a bare -log(symbol) term is supported, got {term}")
            a = _atom_from_sympy(rest.args[0])
            if _atom_value(a) is not None:
                raise NotRepresentable(f"log of the constant {rest.args[0]}")
            out = out + Poly.minus_log(a)
            continue
        p = Poly.const(Fraction(int(c.p), int(c.q)))
        for base, e in rest.as_powers_dict().items():
            if base.is_Number:
                if base != 1:
                    raise NotRepresentable(f"{term}")
                continue
            if not e.is_Integer:
                raise NotRepresentable(f"non-integer power in {term}")
            a = _atom_from_sympy(base)
            v = _atom_value(a)
            if v is not None:
                if v == 0 and int(e) < 0:
                    raise ZeroDivisionError(f"{base} is identically 0")
                p = p * (v ** int(e) if int(e) >= 0 else 1)
            else:
                p = p * Poly.atom(a, int(e))
        out = out + p
    return out


# ---------------------------------------------------------------------------
# compilation to the device table (include/txmom.h: txm_poly_table)
# ---------------------------------------------------------------------------
def compile_table(polys: Iterable[Poly]):
    """-> dict(atoms=[Atom...], func_term0, func_flags, coef, term_fac0, fac_atom, fac_pow, log_atom)
    with plain python lists; atoms are referenced by index."""
    polys = list(polys)
    atoms: list[Atom] = []
    index: dict[Atom, int] = {}

    def aid(a):
        if a not in index:
            index[a] = len(atoms)
            atoms.append(a)
        return index[a]

    func_term0, func_flags, coef, term_fac0, fac_atom, fac_pow = [0], [], [], [0], [], []
    log_atom = -1
    for p in polys:
        flag = 0
        if p.log_atom is not None:
            la = aid(p.log_atom)
            if log_atom not in (-1, la):
                raise NotImplementedError("different log atoms in one table")
            log_atom = la
            flag = 1
        for m, c in sorted(p.terms.items(), key=lambda kv: [(_key(a), q) for a, q in kv[0]]):
            coef.append(float(c))
            for a, q in m:
                fac_atom.append(aid(a))
                fac_pow.append(int(q))
            term_fac0.append(len(fac_atom))
        func_term0.append(len(coef))
        func_flags.append(flag)
    return dict(atoms=atoms, func_term0=func_term0, func_flags=func_flags, coef=coef, term_fac0=term_fac0,
                fac_atom=fac_atom, fac_pow=fac_pow, log_atom=max(log_atom, 0))


# ---------------------------------------------------------------------------
# host evaluation on caller-supplied arguments (reference models.py:317-372: ``funcs[i](*args)``)
# ---------------------------------------------------------------------------
_ROLE = {"W": "u", "xW": "xu"}  # the volume expansion names its raw moments W / xW (reference volume.py:34-60)


def resolve_from_args(names, args):
    """atom -> value for ``Derivatives.derivs(args=...)``: ``names`` are the symbol families in the order the reference
    passes them to its lambdified functions (central: x1, du, dxdu; raw: u, xu; callbacks append their own), ``args``
    the matching objects -- arrays for plain symbols, ``obj[n]`` / ``obj[n, d]`` indexables for the moment families."""
    if len(names) != len(args):
        raise ValueError(f"expected {len(names)} args {tuple(names)}, got {len(args)}")
    by_role = {_ROLE.get(n, n): a for n, a in zip(names, args)}

    def resolve(atom):
        kind = "u" if atom[0] == "umean" else atom[0]
        if kind not in by_role:
            raise ValueError(f"no argument for symbol family {atom[0]!r} (args are {tuple(names)})")
        obj = by_role[kind]
        idx = tuple(v for v in atom[1:] if v is not None)
        if atom[0] == "umean" or not idx:
            return obj
        return obj[idx if len(idx) > 1 else idx[0]]

    return resolve


def eval_host(p: Poly, resolve, absolute: bool = False):
    """sum_t c_t prod atoms^powers (- log(atom)) with ordinary arithmetic on whatever ``resolve(atom)`` returns (numpy
    or labelled arrays, broadcasting as they do).  ``absolute=True`` evaluates sum_t |c_t| prod |atom|^power instead:
    the scale of the first-order rounding-error bound of the evaluation (the condition number of a derivative is this
    over its value; tests hold the device table to 1e-12 of it)."""
    import numpy as np

    cache: dict = {}

    def val(a):
        if a not in cache:
            v = resolve(a)
            cache[a] = abs(v) if absolute else v
        return cache[a]

    out = 0.0
    for m, c in p.terms.items():
        t = abs(float(c)) if absolute else float(c)
        for a, q in m:
            t = t * val(a) ** q
        out = out + t
    if p.log_atom is not None:
        lg = np.log(resolve(p.log_atom))
        out = out + (abs(lg) if absolute else -lg)
    return out


# ---------------------------------------------------------------------------
# sympy view (for symbolic identity tests and `.exprs`)
# ---------------------------------------------------------------------------
def to_sympy(p: Poly):
    """sympy expression over IndexedBase/Symbol names matching the reference's
    (du, dxdu, x1, u, xu; reference beta.py:46, 77, 104, 138, 168, 187, 210, 240)."""
    import sympy as sp

    def sym(a):
        kind = a[0]
        idx = [v for v in a[1:] if v is not None]
        if kind == "umean":
            return sp.Symbol("u")
        if not idx:
            return sp.Symbol(kind)
        return sp.IndexedBase(kind)[tuple(idx) if len(idx) > 1 else idx[0]]

    expr = sp.Integer(0)
    for m, c in p.terms.items():
        t = sp.Rational(c.numerator, c.denominator)
        for a, q in m:
            t = t * sym(a) ** q
        expr = expr + t
    if p.log_atom is not None:
        expr = expr - sp.log(sym(p.log_atom))
    return expr
