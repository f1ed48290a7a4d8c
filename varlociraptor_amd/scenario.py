"""Host-side scenario model: samples, universes, events -> flattened VAF trees (vlr_scenario_desc).

Mirrors the reference's grammar front-end for the part the hot path consumes:
  * grammar::Scenario / Sample (src/grammar/mod.rs:129-144, 471-496), sample index = sorted name
    order (mod.rs:178-190), default resolution 0.01 (mod.rs:445-447);
  * VAFSpectrum / VAFRange syntax of formula.pest:1-4 (`[0.0,0.5[ | 0.5 | 1.0`, `{0.0,0.5}`);
  * VAFTree::new from a *normalized* formula (src/grammar/vaftree.rs:168-305) incl.
    add_missing_samples and the operand ordering of Formula::sort (formula.rs:455-471).
The formula parser here handles atoms, `&`, `|`, parentheses, `true`/`false`, IUPAC variants and
l2fc() terms and negation pushed down to the atoms (formula.rs:717-905); $expressions and the BDD
simplification (formula.rs:473-530) belong to the "next" row of SURVEY §8(f).
"""
from __future__ import annotations

import ctypes as C
import math
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import abi


# ----------------------------------------------------------------------------- spectra
@dataclass(frozen=True)
class VAFRange:
    start: float
    end: float
    left_exclusive: bool
    right_exclusive: bool

    def is_complete(self) -> bool:  # formula.rs:1082-1084
        return self.start == 0.0 and self.end == 1.0 and not self.left_exclusive and not self.right_exclusive


@dataclass(frozen=True)
class VAFSet:
    vafs: Tuple[float, ...]  # ascending (BTreeSet)


Spectrum = Union[VAFRange, VAFSet]

_VAF_RE = r"(?:0\.\d+|1\.0)"  # formula.pest: vaf


def parse_vafdef(text: str) -> Spectrum:
    """Parse one `vafdef` of formula.pest (vaf | vafrange | vafset)."""
    t = text.strip().replace(" ", "")
    m = re.fullmatch(r"([\[\]])(%s),(%s)([\[\]])" % (_VAF_RE, _VAF_RE), t)
    if m:
        return VAFRange(float(m.group(2)), float(m.group(3)), m.group(1) == "]", m.group(4) == "[")
    m = re.fullmatch(r"\{(%s(?:,%s)+)\}" % (_VAF_RE, _VAF_RE), t)
    if m:
        return VAFSet(tuple(sorted(set(float(v) for v in m.group(1).split(",")))))
    if re.fullmatch(_VAF_RE, t):
        return VAFSet((float(t),))
    raise ValueError("invalid VAF definition: %r" % text)


def parse_universe(text: str) -> List[Spectrum]:
    """formula.pest `universe` rule; VAFUniverse is a BTreeSet of spectra (Sets sort before Ranges,
    formula.rs:1018-1022), iteration order matters for add_missing_samples child order."""
    specs = [parse_vafdef(p) for p in text.split("|")]
    uniq = list(dict.fromkeys(specs))

    def key(s):
        if isinstance(s, VAFSet):
            return (0, s.vafs)
        return (1, (s.start, s.end, s.left_exclusive, s.right_exclusive))

    return sorted(uniq, key=key)


# ----------------------------------------------------------------------------- normalized formula AST
@dataclass
class Atom:
    sample: str
    vafs: Spectrum


@dataclass
class Variant:
    refbase: str
    altbase: str
    positive: bool = True


@dataclass
class Lfc:
    sample_a: str
    sample_b: str
    cmp: int
    value: float


@dataclass
class Conj:
    operands: list


@dataclass
class Disj:
    operands: list


@dataclass
class Const:
    value: bool


@dataclass
class Neg:
    operand: object


_CMP = {"==": abi.CMP_EQUAL, ">": abi.CMP_GREATER, ">=": abi.CMP_GREATER_EQUAL, "<": abi.CMP_LESS,
        "<=": abi.CMP_LESS_EQUAL, "!=": abi.CMP_NOT_EQUAL}


@dataclass(frozen=True)
class ExprRef:
    """`$identifier` (formula.pest: expression); expanded from Scenario.expressions before normalisation."""
    identifier: str


class _Parser:
    """Recursive-descent parser for formula.pest."""

    def __init__(self, text: str):
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)  # COMMENT = "/*" ... "*/" (formula.pest)
        self.s = text.replace(" ", "")
        self.i = 0

    def peek(self, n=1):
        return self.s[self.i:self.i + n]

    def parse(self):
        f = self.expr()
        if self.i != len(self.s):
            raise ValueError("trailing input in formula at %d: %r" % (self.i, self.s))
        return f

    def expr(self):
        first = self.sub()
        if self.peek() == "&":
            ops = [first]
            while self.peek() == "&":
                self.i += 1
                ops.append(self.sub())
            return Conj(ops)
        if self.peek() == "|":
            ops = [first]
            while self.peek() == "|":
                self.i += 1
                ops.append(self.sub())
            return Disj(ops)
        return first

    def sub(self):
        if self.peek() == "(":
            self.i += 1
            f = self.expr()
            if self.peek() != ")":
                raise ValueError("missing ) in formula")
            self.i += 1
            return f
        if self.peek() == "!":
            self.i += 1
            return Neg(self.sub())
        if self.peek() == "$":
            m = re.match(r"\$([\w.\-]+)", self.s[self.i:])
            if not m:
                raise ValueError("invalid expression reference at %r" % self.s[self.i:])
            self.i += m.end()
            return ExprRef(m.group(1))
        if self.s.startswith("l2fc(", self.i):
            m = re.match(r"l2fc\(([\w.\-]+),([\w.\-]+)\)(<=|<|>=|>|!=|==)(-?\d+(?:\.\d*)?(?:[eE][+-]?\d+)?)", self.s[self.i:])
            if not m:
                raise ValueError("invalid l2fc term")
            self.i += m.end()
            return Lfc(m.group(1), m.group(2), _CMP[m.group(3)], float(m.group(4)))
        if self.s.startswith("true", self.i):
            self.i += 4
            return Const(True)
        if self.s.startswith("false", self.i):
            self.i += 5
            return Const(False)
        m = re.match(r"([ACGTRYSWKMBDHVN])>([ACGTRYSWKMBDHVN])(?![\w:])", self.s[self.i:])
        if m:
            self.i += m.end()
            return Variant(m.group(1), m.group(2), True)
        # cmp = identifier cmp_ops identifier: a log2 fold change predicate with value 0 (formula.rs:1532-1545)
        m = re.match(r"([\w.\-]+?)(<=|<|>=|>|!=|==)([\w.\-]+)", self.s[self.i:])
        if m and not re.match(r"[\w.\-]+:", self.s[self.i:]):
            self.i += m.end()
            return Lfc(m.group(1), m.group(3), _CMP[m.group(2)], 0.0)
        m = re.match(r"([\w.\-]+):", self.s[self.i:])
        if not m:
            raise ValueError("cannot parse formula at %r" % self.s[self.i:])
        name = m.group(1)
        self.i += m.end()
        m = re.match(r"[\[\]]%s,%s[\[\]]|\{[^}]*\}|%s" % (_VAF_RE, _VAF_RE, _VAF_RE), self.s[self.i:])
        if not m:
            raise ValueError("invalid VAF definition at %r" % self.s[self.i:])
        self.i += m.end()
        return Atom(name, parse_vafdef(m.group(0)))


def parse_formula(text: str):
    return _Parser(text).parse()


# ----------------------------------------------------------------------------- scenario
@dataclass
class Contamination:
    by: str
    fraction: float


@dataclass
class Inheritance:
    kind: int  # abi.INHERIT_*
    parents: Tuple[str, ...] = ()
    somatic: bool = False


@dataclass
class Sample:
    resolution: float = 0.01  # grammar/mod.rs:445-447
    universe: Optional[str] = None
    contamination: Optional[Contamination] = None
    ploidy: Optional[int] = None
    somatic_effective_mutation_rate: Optional[float] = None
    germline_mutation_rate: Optional[float] = None
    inheritance: Optional[Inheritance] = None


@dataclass
class Species:
    heterozygosity: Optional[float] = None
    germline_mutation_rate: Optional[float] = None
    somatic_effective_mutation_rate: Optional[float] = None
    ploidy: Optional[int] = None
    fraction_indel: float = 0.0125  # grammar/mod.rs:386-396
    fraction_mnv: float = 0.001
    fraction_sv: float = 0.01


@dataclass
class _TNode:
    kind: int
    sample: int = 0
    sample_b: int = 0
    cmp: int = 0
    lfc_value: float = 0.0
    vafs: Optional[Spectrum] = None
    positive: bool = False
    refbase: str = "N"
    altbase: str = "N"
    children: List["_TNode"] = field(default_factory=list)

    def clone(self):
        return _TNode(self.kind, self.sample, self.sample_b, self.cmp, self.lfc_value, self.vafs, self.positive,
                      self.refbase, self.altbase, [c.clone() for c in self.children])

    def leafs(self):
        if not self.children:
            return [self]
        out = []
        for c in self.children:
            out.extend(c.leafs())
        return out


class Scenario:
    """grammar::Scenario for one contig.  `events` maps name -> formula string or normalized AST."""

    def __init__(self, samples: Dict[str, Sample], events: Dict[str, object], species: Optional[Species] = None,
                 full_prior: bool = False, expressions: Optional[Dict[str, object]] = None):
        self.samples = dict(sorted(samples.items()))  # BTreeMap order
        self.sample_names = list(self.samples.keys())
        self.idx = {n: i for i, n in enumerate(self.sample_names)}
        if len(self.sample_names) > abi.MAX_SAMPLES:
            raise ValueError("at most %d samples supported" % abi.MAX_SAMPLES)
        self.species = species
        self.full_prior = full_prior
        self.events = dict(sorted(events.items()))  # BTreeMap order (grammar/mod.rs:137)
        self.event_names = list(self.events.keys())
        self.expressions = dict(expressions or {})
        # Scenario::from_path (grammar/mod.rs:147-168): every event is also a reusable expression, and `$absent` is the
        # conjunction of sample:0.0 over all samples (Formula::absent, formula.rs:440-453) unless the user defines one
        for name, formula in self.events.items():
            self.expressions.setdefault(name, formula)
        self.expressions.setdefault("absent", Conj([Atom(n, VAFSet((0.0,))) for n in self.samples]) if len(self.samples) > 1
                                    else Atom(next(iter(self.samples)), VAFSet((0.0,))))
        # per-variant prior overrides (LogProb) taken from the first candidate record of a contig (calling.rs:470-494, 704-713)
        self.variant_heterozygosity_ln = None
        self.variant_somatic_effective_mutation_rate_ln = None
        self._keep = []

    # Sample::contig_ploidy (grammar/mod.rs:581-593)
    def ploidy(self, name: str) -> Optional[int]:
        s = self.samples[name]
        if s.ploidy is not None:
            return s.ploidy
        if self.species is not None:
            return self.species.ploidy
        return None

    def somatic_rate(self, name):  # grammar/mod.rs:606-616
        s = self.samples[name]
        if s.somatic_effective_mutation_rate is not None:
            return s.somatic_effective_mutation_rate
        return self.species.somatic_effective_mutation_rate if self.species else None

    def germline_rate(self, name):  # grammar/mod.rs:594-604
        s = self.samples[name]
        if s.germline_mutation_rate is not None:
            return s.germline_mutation_rate
        return self.species.germline_mutation_rate if self.species else None

    # Sample::contig_universe (grammar/mod.rs:503-579)
    def universe(self, name: str) -> List[Spectrum]:
        s = self.samples[name]
        if s.universe is not None:
            return parse_universe(s.universe)
        ploidy = self.ploidy(name)
        has_somatic = s.somatic_effective_mutation_rate is not None  # NB: sample-level only (mod.rs:535)
        if ploidy is not None:
            pts = tuple(sorted(set((n / ploidy if ploidy > 0 else 0.0) for n in range(ploidy + 1))))
            if not has_somatic:
                return [VAFSet(pts)]
            specs: List[Spectrum] = [VAFSet(pts)]
            for a, b in zip(pts[:-1], pts[1:]):
                specs.append(VAFRange(a, b, True, True))
            return specs  # Sets sort before Ranges in the BTreeSet
        if has_somatic:
            return [VAFRange(0.0, 1.0, False, False)]
        raise ValueError("sample needs to define either universe, ploidy or somatic_mutation_rate")

    # ---- VAFTree::new (grammar/vaftree.rs:168-305)
    def _sorted_operands(self, ops):
        # Formula::sort (formula.rs:455-471): derive(Ord) => Conjunction < Disjunction < Terminal;
        # atoms by (sample, vafs); then LFC terms first (stable).
        def rank(o):
            if isinstance(o, Conj):
                return (0, "")
            if isinstance(o, Disj):
                return (1, "")
            if isinstance(o, Atom):
                return (2, o.sample)
            if isinstance(o, Variant):
                return (3, "")
            if isinstance(o, Lfc):
                return (4, "")
            return (5, "")  # False < True (formula.rs:103-124 FormulaTerminal order)
        s = sorted(ops, key=rank)
        return sorted(s, key=lambda o: 0 if isinstance(o, Lfc) else 1)

    def _from(self, f) -> List[_TNode]:
        if isinstance(f, Atom):
            if f.sample not in self.idx:
                raise ValueError("invalid sample name %r" % f.sample)
            return [_TNode(abi.NODE_SAMPLE, sample=self.idx[f.sample], vafs=f.vafs)]
        if isinstance(f, Disj):
            out = []
            for o in self._sorted_operands(f.operands):
                out.extend(self._from(o))
            return out
        if isinstance(f, Conj):
            ops = self._sorted_operands(f.operands)
            ops = sorted(ops, key=lambda o: 1 if isinstance(o, Disj) else 0)  # disjunctions to the end (vaftree.rs:199-206)
            roots = self._from(ops[0])
            for o in ops[1:]:
                subtrees = self._from(o)
                for r in roots:
                    for leaf in r.leafs():
                        leaf.children = [t.clone() for t in subtrees]
            return roots
        if isinstance(f, Variant):
            return [_TNode(abi.NODE_VARIANT, positive=f.positive, refbase=f.refbase, altbase=f.altbase)]
        if isinstance(f, Const):
            return [_TNode(abi.NODE_TRUE if f.value else abi.NODE_FALSE)]
        if isinstance(f, Lfc):
            return [_TNode(abi.NODE_LFC, sample=self.idx[f.sample_a], sample_b=self.idx[f.sample_b], cmp=f.cmp,
                           lfc_value=f.value)]
        raise TypeError(f)

    def _add_missing(self, node: _TNode, seen: set):
        if node.kind == abi.NODE_FALSE:
            return
        if node.kind == abi.NODE_SAMPLE:
            seen.add(node.sample)
        if not node.children:
            for name in self.sample_names:
                i = self.idx[name]
                if i not in seen:
                    seen.add(i)
                    node.children = [_TNode(abi.NODE_SAMPLE, sample=i, vafs=sp) for sp in self.universe(name)]
                    self._add_missing(node, seen)
                    break
        else:
            if len(node.children) > 1:
                for ch in node.children[1:]:
                    self._add_missing(ch, set(seen))
            self._add_missing(node.children[0], seen)

    # ---- Formula::negate / apply_negations (grammar/formula.rs:717-905), without the BDD simplification
    def _split_at(self, r: VAFRange, vaf: float):
        """VAFRange::split_at (formula.rs:1099-1129)."""
        def to_spec(start, end, lex, rex):
            if start == end:
                if not (lex and r.right_exclusive):
                    return VAFSet((start,))
                return None
            return VAFRange(start, end, lex, rex)
        return to_spec(r.start, vaf, r.left_exclusive, True), to_spec(vaf, r.end, True, r.right_exclusive)

    @staticmethod
    def _overlap(a: VAFRange, b: VAFRange) -> str:
        """VAFRange::overlap (formula.rs:1131-1168) of a relative to b."""
        if a == b:
            return "Equal"
        start_right = (a.start >= b.start) if (a.left_exclusive and not b.left_exclusive) else (a.start > b.start)
        end_left = (a.end <= b.end) if (a.right_exclusive and not b.right_exclusive) else (a.end < b.end)
        if (a.end < b.start or a.start > b.end) or (a.end <= b.start and (a.right_exclusive or b.left_exclusive)) or \
                (a.start >= b.end and (a.left_exclusive or b.right_exclusive)):
            return "None"
        return {(True, True): "Contained", (True, False): "Start", (False, True): "End", (False, False): "Contains"}[(start_right, end_left)]

    @staticmethod
    def _contains(r: VAFRange, v: float) -> bool:
        lo = r.start < v if r.left_exclusive else r.start <= v
        hi = r.end > v if r.right_exclusive else r.end >= v
        return lo and hi

    def _negate(self, f):
        if isinstance(f, Const):
            return Const(not f.value)
        if isinstance(f, Conj):
            return Disj([self._negate(o) for o in f.operands])
        if isinstance(f, Disj):
            return Conj([self._negate(o) for o in f.operands])
        if isinstance(f, Neg):
            return self._apply_negations(f.operand)
        if isinstance(f, Variant):
            return Variant(f.refbase, f.altbase, not f.positive)
        if isinstance(f, Lfc):
            inv = {abi.CMP_EQUAL: abi.CMP_NOT_EQUAL, abi.CMP_GREATER: abi.CMP_LESS_EQUAL, abi.CMP_GREATER_EQUAL: abi.CMP_LESS,
                   abi.CMP_LESS: abi.CMP_GREATER_EQUAL, abi.CMP_LESS_EQUAL: abi.CMP_GREATER, abi.CMP_NOT_EQUAL: abi.CMP_EQUAL}
            return Lfc(f.sample_a, f.sample_b, inv[f.cmp], f.value)  # utils/comparison.rs:28-41
        assert isinstance(f, Atom)
        universe = self.universe(f.sample)
        out: List[Spectrum] = []
        if isinstance(f.vafs, VAFSet):
            stack = list(universe)
            while stack:
                u = stack.pop(0)
                if isinstance(u, VAFSet):
                    diff = tuple(v for v in u.vafs if v not in f.vafs.vafs)
                    if diff:
                        out.append(VAFSet(diff))
                else:
                    for vaf in f.vafs.vafs:
                        if self._contains(u, vaf):
                            left, right = self._split_at(u, vaf)
                            if right is not None:
                                stack.append(right)
                            if left is not None:
                                out.append(left)
                        else:
                            out.append(u)
        else:
            rng = f.vafs
            for u in universe:
                if isinstance(u, VAFSet):
                    keep = tuple(v for v in u.vafs if not self._contains(rng, v))
                    if keep:
                        out.append(VAFSet(keep))
                else:
                    ov = self._overlap(rng, u)
                    if ov == "Contained":
                        l = self._split_at(u, rng.start)[0]
                        r = self._split_at(u, rng.end)[1]
                        out.extend(x for x in (l, r) if x is not None)
                    elif ov == "End":
                        r = self._split_at(u, rng.end)[1]
                        if r is not None:
                            out.append(r)
                    elif ov == "Start":
                        l = self._split_at(u, rng.start)[0]
                        if l is not None:
                            out.append(l)
                    elif ov == "None":
                        out.append(u)
        if not out:
            return Atom(f.sample, VAFSet(()))
        return Disj([Atom(f.sample, sp) for sp in out])

    def _apply_negations(self, f):
        if isinstance(f, Neg):
            return self._negate(self._apply_negations(f.operand)) if isinstance(f.operand, Neg) else self._negate(f.operand)
        if isinstance(f, Conj):
            return Conj([self._apply_negations(o) for o in f.operands])
        if isinstance(f, Disj):
            return Disj([self._apply_negations(o) for o in f.operands])
        return f

    @staticmethod
    def _flatten(f):
        """Minimal stand-in for Formula::simplify: merge nested conjunctions/disjunctions, unwrap singletons."""
        if isinstance(f, (Conj, Disj)):
            ops = []
            for o in f.operands:
                o = Scenario._flatten(o)
                if type(o) is type(f):
                    ops.extend(o.operands)
                else:
                    ops.append(o)
            if len(ops) == 1:
                return ops[0]
            return type(f)(ops)
        return f

    # ---- Formula::normalize (grammar/formula.rs:473-485): expand expressions, push negations into the atoms, simplify,
    # merge the atoms of one sample, simplify again, strip false, sort.  `simplify` stands in for the reference's
    # BDD round trip (boolean_expression::Expr::simplify_via_bdd, third-party): after apply_negations every formula is
    # monotone in its terminals, and the minimal cover of a monotone function is its set of prime implicants, i.e.
    # distribution into a disjunction of conjunctions plus absorption.
    def _expand(self, f, depth=0):
        if depth > 64:
            raise ValueError("cyclic scenario expressions")
        if isinstance(f, ExprRef):
            if f.identifier not in self.expressions:
                raise ValueError("undefined expression %r" % f.identifier)  # errors::Error::UndefinedExpression
            e = self.expressions[f.identifier]
            if isinstance(e, str):
                e = parse_formula(e)
            return self._expand(e, depth + 1)
        if isinstance(f, Conj):
            return Conj([self._expand(o, depth) for o in f.operands])
        if isinstance(f, Disj):
            return Disj([self._expand(o, depth) for o in f.operands])
        if isinstance(f, Neg):
            return Neg(self._expand(f.operand, depth))
        return f

    @staticmethod
    def _lit_key(o):
        if isinstance(o, Atom):
            v = o.vafs
            return ("atom", o.sample, ("set",) + tuple(v.vafs) if isinstance(v, VAFSet) else ("range", v.start, v.end, v.left_exclusive, v.right_exclusive))
        if isinstance(o, Variant):
            return ("variant", o.positive, o.refbase, o.altbase)
        if isinstance(o, Lfc):
            return ("lfc", o.sample_a, o.sample_b, o.cmp, o.value)
        raise TypeError(o)

    @classmethod
    def _cubes(cls, f) -> List[List[object]]:
        """Disjunction of conjunctions of terminals: [] = false, [[]] = true."""
        if isinstance(f, Const):
            return [[]] if f.value else []
        if isinstance(f, Disj):
            out = []
            for o in f.operands:
                out.extend(cls._cubes(o))
            return out
        if isinstance(f, Conj):
            acc = [[]]
            for o in f.operands:
                nxt = []
                for a in acc:
                    for b in cls._cubes(o):
                        cube, seen = [], set()
                        for lit in a + b:
                            k = cls._lit_key(lit)
                            if k not in seen:
                                seen.add(k)
                                cube.append(lit)
                        nxt.append(cube)
                acc = nxt
            return acc
        return [[f]]

    @classmethod
    def _simplify(cls, f):
        cubes = cls._cubes(f)
        keyed = [(frozenset(cls._lit_key(l) for l in c), c) for c in cubes]
        out = []
        for i, (ki, ci) in enumerate(keyed):
            absorbed = False
            for j, (kj, _) in enumerate(keyed):
                if i != j and (kj < ki or (kj == ki and j < i)):  # a proper sub-cube (or an earlier duplicate) covers this one
                    absorbed = True
                    break
            if not absorbed:
                out.append(ci)
        if not out:
            return Const(False)
        if any(len(c) == 0 for c in out):
            return Const(True)
        terms = [c[0] if len(c) == 1 else Conj(list(c)) for c in out]
        return terms[0] if len(terms) == 1 else Disj(terms)

    @classmethod
    def _range_and(cls, a: VAFRange, b: VAFRange) -> VAFRange:  # `&` (formula.rs:1268-1286)
        ov = cls._overlap(a, b)
        if ov in ("Contained", "Equal"):
            return a
        if ov == "Contains":
            return b
        if ov == "Start":
            return VAFRange(a.start, b.end, a.left_exclusive, b.right_exclusive)
        if ov == "End":
            return VAFRange(b.start, a.end, b.left_exclusive, a.right_exclusive)
        return VAFRange(0.0, 0.0, True, True)

    @classmethod
    def _range_or(cls, a: VAFRange, b: VAFRange) -> VAFRange:  # `|` (formula.rs:1288-1306), overlapping operands only
        ov = cls._overlap(a, b)
        if ov == "Contained":
            return b
        if ov in ("Contains", "Equal"):
            return a
        if ov == "Start":
            return VAFRange(b.start, a.end, b.left_exclusive, a.right_exclusive)
        return VAFRange(a.start, b.end, a.left_exclusive, b.right_exclusive)  # End

    @staticmethod
    def _spec_empty(v: Spectrum) -> bool:
        if isinstance(v, VAFSet):
            return len(v.vafs) == 0
        return v.start == v.end and (v.left_exclusive or v.right_exclusive)

    @classmethod
    def _merge_conj(cls, a: Spectrum, b: Spectrum) -> Spectrum:  # FormulaTerminal::merge_conjunctions (127-168)
        if isinstance(a, VAFRange) and isinstance(b, VAFRange):
            r = cls._range_and(a, b)
            if r.start == r.end and not (r.left_exclusive or r.right_exclusive):
                return VAFSet((a.start,))
            return r
        if isinstance(a, VAFRange):
            return VAFSet(tuple(v for v in b.vafs if cls._contains(a, v)))
        if isinstance(b, VAFRange):
            return VAFSet(tuple(v for v in a.vafs if cls._contains(b, v)))
        return VAFSet(tuple(v for v in a.vafs if v in b.vafs))

    @classmethod
    def _try_merge_disj(cls, a: Spectrum, b: Spectrum) -> Optional[Spectrum]:  # try_merge_disjunction (170-210)
        if isinstance(a, VAFRange) and isinstance(b, VAFRange):
            return None if cls._overlap(a, b) == "None" else cls._range_or(a, b)
        if isinstance(a, VAFRange):
            return a if all(cls._contains(a, v) for v in b.vafs) else None
        if isinstance(b, VAFRange):
            return b if all(cls._contains(b, v) for v in a.vafs) else None
        return VAFSet(tuple(sorted(set(a.vafs) | set(b.vafs))))

    @classmethod
    def _merge_atoms(cls, f):  # formula.rs:575-689
        if isinstance(f, Conj):
            groups: Dict[Optional[str], list] = {}
            for o in f.operands:
                groups.setdefault(o.sample if isinstance(o, Atom) else None, []).append(o)
            ops = []
            for sample, stmts in groups.items():
                if sample is None:
                    ops.extend(cls._merge_atoms(o) for o in stmts)
                    continue
                merged = stmts[-1].vafs
                for o in stmts[:-1]:
                    merged = cls._merge_conj(merged, o.vafs)
                if cls._spec_empty(merged):
                    return Const(False)
                ops.append(Atom(sample, merged))
            return Conj(ops)
        if isinstance(f, Disj):
            groups = {}
            for o in f.operands:
                groups.setdefault(o.sample if isinstance(o, Atom) else None, []).append(o)
            ops = []
            for sample, stmts in groups.items():
                if sample is None:
                    ops.extend(cls._merge_atoms(o) for o in stmts)
                    continue
                # the reference sorts by start only, with an UNSTABLE sort (formula.rs:636-648): the order among equal starts
                # is unspecified there and decides how far the greedy merge gets.  Ties are broken here so that the result
                # does not depend on the operand order: ranges before sets, inclusive before exclusive starts, longer first.
                def merge_key(a):
                    v = a.vafs
                    if isinstance(v, VAFSet):
                        return (min(v.vafs) if v.vafs else -1.0, 1, 0, 0.0)
                    return (v.start, 0, int(v.left_exclusive), -v.end)
                stmts = sorted(stmts, key=merge_key)
                cur = stmts[0].vafs
                for o in stmts[1:]:
                    m = cls._try_merge_disj(cur, o.vafs)
                    if m is not None:
                        cur = m
                    else:
                        ops.append(Atom(sample, cur))
                        cur = o.vafs
                ops.append(Atom(sample, cur))
            return Disj(ops)
        if isinstance(f, Neg):
            return Neg(cls._merge_atoms(f.operand))
        return f

    @staticmethod
    def _strip_false(f):  # formula.rs:691-707
        if isinstance(f, Disj):
            keep = []
            for o in f.operands:
                if isinstance(o, Const) and not o.value:
                    continue
                if isinstance(o, Conj) and any(isinstance(x, Const) and not x.value for x in o.operands):
                    continue
                keep.append(o)
            return Disj(keep)
        return f

    def normalize(self, f):
        if isinstance(f, str):
            f = parse_formula(f)
        f = self._apply_negations(self._expand(f))
        f = self._simplify(self._merge_atoms(self._simplify(f)))
        return self._strip_false(f)

    def canonical(self, f) -> tuple:
        """Order-independent rendering of a normalised formula (what Formula::sort + derive(Eq) compare)."""
        f = self.normalize(f)

        def canon(o):
            if isinstance(o, Conj):
                return ("and", tuple(sorted(canon(x) for x in o.operands)))
            if isinstance(o, Disj):
                return ("or", tuple(sorted(canon(x) for x in o.operands)))
            if isinstance(o, Const):
                return ("const", o.value)
            return self._lit_key(o)
        return canon(f)

    def validate(self):
        """Scenario::validate (grammar/mod.rs:223-279): two events whose disjunction is itself one of the events overlap
        (errors::Error::OverlappingEvents).  The reference runs this from `vaftrees()` for every contig."""
        names = {}
        for name, formula in self.events.items():
            if name == "absent":
                continue
            names.setdefault(self.canonical(formula), []).append(name)
        keys = sorted(names, key=repr)
        overlapping = []
        for i, e1 in enumerate(keys):
            for e2 in keys[i + 1:]:
                if e1 == ("const", False) or e2 == ("const", False):
                    continue
                f1, f2 = self.events[names[e1][0]], self.events[names[e2][0]]
                f1 = parse_formula(f1) if isinstance(f1, str) else f1
                f2 = parse_formula(f2) if isinstance(f2, str) else f2
                d = self.canonical(Disj([f1, f2]))
                if d in names:
                    overlapping.append("(%r | %r) = %r" % (names[e1], names[e2], names[d]))
        if overlapping:
            raise ValueError("overlapping events: " + ", ".join(overlapping))

    def vaftree(self, event: str) -> List[_TNode]:
        roots = self._from(self.normalize(self.events[event]))
        for r in roots:
            self._add_missing(r, set())
        return roots

    # ---- flatten to vlr_scenario_desc
    def desc(self) -> abi.ScenarioDesc:
        S = len(self.sample_names)
        vafs_pool: List[float] = []

        def spec_struct(sp: Spectrum) -> abi.Spectrum:
            st = abi.Spectrum()
            if isinstance(sp, VAFSet):
                st.kind = abi.SPECTRUM_SET
                st.set_offset = len(vafs_pool)
                st.set_len = len(sp.vafs)
                vafs_pool.extend(sp.vafs)
            else:
                st.kind = abi.SPECTRUM_RANGE
                st.start, st.end = sp.start, sp.end
                st.left_exclusive, st.right_exclusive = int(sp.left_exclusive), int(sp.right_exclusive)
            return st

        nodes: List[abi.Node] = []
        child_index: List[int] = []
        root_index: List[int] = []
        root_offset = [0]

        def emit(n: _TNode) -> int:
            my = len(nodes)
            st = abi.Node()
            nodes.append(st)
            st.kind, st.sample, st.sample_b, st.cmp, st.lfc_value = n.kind, n.sample, n.sample_b, n.cmp, n.lfc_value
            if n.kind == abi.NODE_SAMPLE:
                st.vafs = spec_struct(n.vafs)
            st.positive = int(n.positive)
            st.refbase, st.altbase = ord(n.refbase), ord(n.altbase)
            ids = [emit(c) for c in n.children]
            st.child_offset = len(child_index)
            st.n_children = len(ids)
            child_index.extend(ids)
            return my

        for ev in self.event_names:
            for r in self.vaftree(ev):
                root_index.append(emit(r))
            root_offset.append(len(root_index))

        uni_off = [0]
        uni: List[abi.Spectrum] = []
        for name in self.sample_names:
            for sp in self.universe(name):
                uni.append(spec_struct(sp))
            uni_off.append(len(uni))

        d = abi.ScenarioDesc()
        keep = self._keep = []

        def arr(ctype, values):
            a = (ctype * max(1, len(values)))(*values)
            keep.append(a)
            return a

        d.n_samples = S
        d.resolution = arr(C.c_double, [self.samples[n].resolution for n in self.sample_names])
        cont_by, cont_fr = [], []
        for n in self.sample_names:
            c = self.samples[n].contamination
            cont_by.append(self.idx[c.by] if c else -1)
            cont_fr.append(c.fraction if c else 0.0)
        d.contaminated_by = arr(C.c_int32, cont_by)
        d.contamination_fraction = arr(C.c_double, cont_fr)
        d.universe_offset = arr(C.c_int32, uni_off)
        d.universe = arr(abi.Spectrum, uni)
        d.uniform_prior = arr(C.c_uint8, [1 if self.samples[n].universe is not None else 0 for n in self.sample_names])
        d.ploidy = arr(C.c_int32, [(-1 if self.ploidy(n) is None else self.ploidy(n)) for n in self.sample_names])
        nan = float("nan")
        d.germline_mutation_rate = arr(C.c_double, [nan if self.germline_rate(n) is None else self.germline_rate(n) for n in self.sample_names])
        d.somatic_effective_mutation_rate = arr(C.c_double, [nan if self.somatic_rate(n) is None else self.somatic_rate(n) for n in self.sample_names])
        inh = []
        for n in self.sample_names:
            st = abi.Inheritance()
            i = self.samples[n].inheritance
            if i is None:
                st.kind, st.from0, st.from1, st.somatic = abi.INHERIT_NONE, -1, -1, 0
            else:
                st.kind = i.kind
                st.from0 = self.idx[i.parents[0]]
                st.from1 = self.idx[i.parents[1]] if len(i.parents) > 1 else -1
                st.somatic = int(i.somatic)
            inh.append(st)
        d.inheritance = arr(abi.Inheritance, inh)
        sp = self.species
        d.heterozygosity = nan if (sp is None or sp.heterozygosity is None) else sp.heterozygosity
        d.variant_heterozygosity_ln = nan if self.variant_heterozygosity_ln is None else float(self.variant_heterozygosity_ln)
        d.variant_somatic_effective_mutation_rate_ln = (nan if self.variant_somatic_effective_mutation_rate_ln is None
                                                        else float(self.variant_somatic_effective_mutation_rate_ln))
        d.fraction_indel = sp.fraction_indel if sp else 0.0125
        d.fraction_mnv = sp.fraction_mnv if sp else 0.001
        d.fraction_sv = sp.fraction_sv if sp else 0.01
        d.is_absent_only = int(not self.full_prior)
        d.n_events = len(self.event_names)
        d.event_names = arr(C.c_char_p, [e.encode() for e in self.event_names])
        d.event_root_offset = arr(C.c_int32, root_offset)
        d.root_index = arr(C.c_int32, root_index)
        d.n_nodes = len(nodes)
        d.nodes = arr(abi.Node, nodes)
        d.child_index = arr(C.c_int32, child_index)
        d.vafs = arr(C.c_double, vafs_pool)
        return d

    @property
    def n_out(self) -> int:
        return len(self.event_names) + 2

    def out_names(self) -> List[str]:
        return ["absent"] + self.event_names + ["artifact"]


# ----------------------------------------------------------------------------- canned scenarios
def tumor_normal(purity: float = 0.75) -> Scenario:
    """The embedded scenario of `call variants tumor-normal` (src/cli.rs:1151-1172)."""
    samples = {
        "tumor": Sample(resolution=0.01, universe="[0.0,1.0]", contamination=Contamination("normal", 1.0 - purity)),
        "normal": Sample(resolution=0.1, universe="[0.0,0.5[ | 0.5 | 1.0"),
    }
    events = {
        "somatic_tumor": "tumor:]0.0,1.0] & normal:0.0",
        "somatic_normal": "tumor:]0.0,1.0] & normal:]0.0,0.5[",
        "germline_het": "tumor:]0.0,1.0] & normal:0.5",
        "germline_hom": "tumor:]0.0,1.0] & normal:1.0",
    }
    return Scenario(samples, events)


def single_sample(resolution: float = 0.01, name: str = "s", event: str = "present") -> Scenario:
    """BASELINE config 2 / the flamegraph_profiling fixture shape: one sample, universe [0,1],
    event `present: s:]0.0,1.0]` (tests/resources/flamegraph_profiling/scenario.yaml)."""
    return Scenario({name: Sample(resolution=resolution, universe="[0.0,1.0]")}, {event: "%s:]0.0,1.0]" % name})
