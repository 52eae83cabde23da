"""gointerp -- executes the reference's OWN Go sources in an image that has no Go toolchain.

A tree-walking interpreter for the subset of Go that go-tfhe's hot path is written in (structs, slices with shared backing
arrays, pointer-to-array views through unsafe.Pointer, value semantics of structs and arrays, methods, closures, multiple
results, switch, defer, complex128, fixed-width unsigned arithmetic that wraps).  It knows nothing about TFHE: it parses the
files under /root/reference with the parser of gocheck.py (function bodies on first call) and runs them.

Why it exists: SURVEY.md 8(c) / DESIGN.md section 4 -- the C oracle is a hand-written RESTATEMENT of the reference, and the
reference itself (pure Go) could not be run here, so ciphertext-level parity was "unpinned".  With this, the reference's
functions -- poly.Evaluator.ToFourierPolyAssign, Evaluator.ExternalProductAssign / CMuxAssign / BlindRotateAssign /
BootstrapAssign / BootstrapLUTAssign, trgsw.IdentityKeySwitchingAssign, gates.*, lut.Generator.GenLookUpTableAssign,
cloudkey.NewCloudKey ... -- are executed from their source text, and their inputs and outputs become fixtures
(tests/golden/goref/, written by tools/go_static/make_goref_vectors.py) that the oracle and the HIP engine are held to.
It is NOT the Go toolchain: the executor is this file (~10^5 Go statements per second: a full-size bootstrap is ~25 minutes, so only a
handful are run at n = 700 and the reference's own key generation is run at a reduced LWE dimension), floating point is IEEE double exactly as Go on amd64 evaluates it (no
fused multiply-add), and math/cmplx functions come from the C library (twiddles within 1 ulp of Go's pure-Go versions:
immaterial where the transforms are exact, inside the stated tolerance elsewhere).

Value model
  int, int64, uint64, uintptr, untyped constants   Python int (no overflow occurs on these paths)
  uint32 (params.Torus), int32, uint8, uint16 ...   numpy scalars: wrap-around arithmetic, C-style conversions
  float64 / complex128 / bool / string              Python float / complex / bool / str (code points, not bytes: len of a non-ASCII
                                                    string differs from Go's; only the examples' banners hold such text)
  struct value                                      GoStruct (copied on assignment, parameter passing, return, element store)
  *T                                                GoPtr(target) -- field access and method calls dereference automatically
  []T                                               GoSlice(backing list, offset, len, cap): slicing SHARES the backing list
  [N]T                                              GoArray (a value); (*[N]T)(unsafe.Pointer(&s[i])) is an ArrayView onto s's backing list
  map / func                                        dict / GoFunc (closures capture variables by reference, as Go does)
Goroutines run inline at the `go` statement (sync.WaitGroup / Mutex are no-ops, sync.Pool.Get calls New): deterministic.
math/rand is backed by a seeded numpy generator -- the reference seeds from the wall clock, so its own runs are not
reproducible either; every random value it draws ends up in the dumped keys and inputs.
"""
import cmath
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gocheck  # noqa: E402

np.seterr(over="ignore", under="ignore")
Node = gocheck.Node


class GoPanic(Exception):
    pass


class TestFatal(Exception):
    """t.Fatal / t.Fatalf / t.FailNow / t.Skip inside a reference test function."""

    def __init__(self, skipped=False):
        super().__init__("test stopped")
        self.skipped = skipped


class Unsupported(Exception):
    pass


# ------------------------------------------------------------------------------------------------------- runtime types

class RT:
    """A resolved Go type."""
    __slots__ = ("kind", "name", "elem", "key", "n", "fields", "under", "pkg", "_zero", "_stub")

    def __init__(self, kind, name=None, elem=None, key=None, n=0, fields=None, under=None, pkg=None):
        self.kind, self.name, self.elem, self.key, self.n, self.fields, self.under, self.pkg = kind, name, elem, key, n, fields, under, pkg

    def u(self):
        t = self
        while t.kind == "named":
            t = t.under()
        return t

    def __repr__(self):
        return f"RT({self.kind} {self.name or ''})"


NP = {"uint32": np.uint32, "int32": np.int32, "uint8": np.uint8, "int8": np.int8, "uint16": np.uint16, "int16": np.int16, "float32": np.float32}
PYINT = {"int", "int64", "uint64", "uintptr", "uint"}
BASIC_RT = {}
for _n in list(NP) + list(PYINT) + ["float64", "complex128", "bool", "string"]:
    BASIC_RT[_n] = RT("basic", _n)
BASIC_RT["byte"] = BASIC_RT["uint8"]
BASIC_RT["rune"] = BASIC_RT["int32"]
RT_IFACE = RT("iface")
RT_ANY = RT("typeparam")
RT_UPTR = RT("basic", "unsafe.Pointer")


class GoStruct:
    __slots__ = ("t", "f")

    def __init__(self, t, f):
        self.t, self.f = t, f


class GoPtr:
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = v


class ElemPtr:
    """&s[i]: a pointer to one element of a backing list."""
    __slots__ = ("a", "i", "et")

    def __init__(self, a, i, et):
        self.a, self.i, self.et = a, i, et


class VarPtr:
    """&x of a local / package variable holding a non-struct value."""
    __slots__ = ("env", "name")

    def __init__(self, env, name):
        self.env, self.name = env, name


class FieldPtr:
    """&x.f of a struct field holding a non-struct value (e.g. &k.ctx handed to C, &s.next handed to atomic.AddUint32)."""
    __slots__ = ("s", "name")

    def __init__(self, s, name):
        self.s, self.name = s, name


class GoSlice:
    __slots__ = ("a", "o", "n", "c", "et")

    def __init__(self, a, o, n, c, et):
        self.a, self.o, self.n, self.c, self.et = a, o, n, c, et


class GoArray:
    """[N]T as a value; also the target of a pointer-to-array view (fixed = True: a window onto someone else's backing list)."""
    __slots__ = ("a", "o", "n", "et")

    def __init__(self, a, o, n, et):
        self.a, self.o, self.n, self.et = a, o, n, et


class GoFunc:
    __slots__ = ("decl", "env", "pkg", "recv", "name")

    def __init__(self, decl, env, pkg, recv=None, name=None):
        self.decl, self.env, self.pkg, self.recv, self.name = decl, env, pkg, recv, name


class Builtin:
    __slots__ = ("fn", "name")

    def __init__(self, fn, name):
        self.fn, self.name = fn, name


class TypeVal:
    __slots__ = ("rt",)

    def __init__(self, rt):
        self.rt = rt


class PkgRef:
    __slots__ = ("pkg",)

    def __init__(self, pkg):
        self.pkg = pkg


class Env:
    __slots__ = ("vars", "parent")

    def __init__(self, parent=None):
        self.vars, self.parent = {}, parent

    def find(self, name):
        e = self
        while e is not None:
            if name in e.vars:
                return e
            e = e.parent
        return None


class Frame:
    __slots__ = ("defers", "results", "named")

    def __init__(self):
        self.defers, self.results, self.named = [], None, None


BREAK, CONTINUE, RETURN = "break", "continue", "return"


def go_round(x):
    """math.Round: nearest integer, halves away from zero (exact for every double)."""
    if x != x or x in (math.inf, -math.inf):
        return x
    a = abs(x)
    if a >= 4503599627370496.0:
        return x
    r = math.floor(a)
    if a - r >= 0.5:
        r += 1.0
    return math.copysign(r, x)


def trunc_div(a, b):
    if b == 0:
        raise GoPanic("integer divide by zero")
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


# ------------------------------------------------------------------------------------------------------- packages

class Pkg:
    def __init__(self, path, name):
        self.path, self.name = path, name
        self.files = []              # (ast, imports)
        self.funcs, self.methods, self.types, self.values = {}, {}, {}, {}
        self.value_decls = {}        # name -> (spec node, index, file ast)
        self.consts = set()
        self.native = {}             # natively implemented members (the standard-library stubs)
        self.env = Env()
        self.rt_cache = {}
        self.initialised = False


class Interp:
    def __init__(self, reference_root, module="github.com/thedonutfactory/go-tfhe", seed=0x7F4E0050):
        self.root, self.module = reference_root, module
        self.pkgs = {}
        self.rng = np.random.RandomState(seed & 0x7FFFFFFF)
        self.steps = 0
        self._std()

    # ---- loading
    def load(self, name):
        path = f"{self.module}/{name}"
        if path in self.pkgs:
            return self.pkgs[path]
        d = os.path.join(self.root, name)
        pkg = Pkg(path, name)
        self.pkgs[path] = pkg
        for fname, src in gocheck.read_dir(d):
            self._add_file(pkg, gocheck.parse_source(src, fname, bodies=False))
        return pkg

    def load_source(self, name, sources, path=None):
        """A package from in-memory sources ({file name: text}): the interpreter's own semantics tests run Go snippets through this."""
        path = path or f"{self.module}/{name}"
        pkg = Pkg(path, name)
        self.pkgs[path] = pkg
        for fname, src in sources.items():
            self._add_file(pkg, gocheck.parse_source(src, fname, bodies=False))
        return pkg

    def _add_file(self, pkg, ast):
        imports = gocheck.World.import_map(ast)
        pkg.files.append((ast, imports))
        for dnode in ast.decls:
            dnode._file = (ast, imports)
            if dnode.kind == "funcdecl":
                if dnode.recv is None:
                    pkg.funcs[dnode.name] = dnode
                else:
                    rt = dnode.recv.type
                    base = rt.elem if rt.kind == "tptr" else rt
                    pkg.methods[(base.name, dnode.name)] = dnode
            elif dnode.kind == "typedecl":
                pkg.types[dnode.name] = dnode
            elif dnode.kind in ("var", "const"):
                for i, n in enumerate(dnode.names):
                    pkg.value_decls[n] = (dnode, i)
                    if dnode.kind == "const":
                        pkg.consts.add(n)

    def pkg_by_import(self, path):
        if path in self.pkgs:
            return self.pkgs[path]
        if path.startswith(self.module + "/"):
            return self.load(path[len(self.module) + 1:])
        for prefix, root in getattr(self, "extra_roots", {}).items():       # e.g. the shim's module: its packages live under shim/go/
            if path.startswith(prefix + "/"):
                return self.load_dir(path, os.path.join(root, path[len(prefix) + 1:]))
        raise Unsupported(f"package {path!r} is not available to the interpreter")

    def load_dir(self, path, d):
        if path in self.pkgs:
            return self.pkgs[path]
        pkg = Pkg(path, os.path.basename(d))
        self.pkgs[path] = pkg
        for fname, src in gocheck.read_dir(d):
            self._add_file(pkg, gocheck.parse_source(src, fname, bodies=False))
        return pkg

    def ensure_init(self, pkg):
        if pkg.initialised:
            return
        pkg.initialised = True
        if "init" in pkg.funcs:
            self.call_decl(pkg.funcs["init"], pkg, [], None)

    # ---- types
    def rtype(self, node, pkg, imports):
        k = node.kind
        if k == "tname":
            if node.pkg is None:
                if node.name in BASIC_RT:
                    return BASIC_RT[node.name]
                if node.name in ("any",):
                    return RT_IFACE
                if node.name == "error":
                    return RT_IFACE
                if node.name in pkg.types or node.name in pkg.native:
                    return self.named(pkg, node.name)
                return RT_ANY                         # a type parameter of a generic function
            target = self.pkg_by_import(imports[node.pkg])
            if target.path == "unsafe" and node.name == "Pointer":
                return RT_UPTR
            return self.named(target, node.name)
        if k == "tptr":
            return RT("ptr", elem=self.rtype(node.elem, pkg, imports))
        if k == "tslice":
            return RT("slice", elem=self.rtype(node.elem, pkg, imports))
        if k == "tarray":
            return RT("array", n=node.len, elem=self.rtype(node.elem, pkg, imports))
        if k == "tmap":
            return RT("map", key=self.rtype(node.key, pkg, imports), elem=self.rtype(node.elem, pkg, imports))
        if k == "functype":
            return RT("func")
        if k == "tiface":
            return RT_IFACE
        if k == "tstruct":
            return RT("struct", fields=[(n, self.rtype(t, pkg, imports)) for n, t in node.fields])
        raise Unsupported(f"type syntax {k}")

    def named(self, pkg, name):
        key = name
        if key in pkg.rt_cache:
            return pkg.rt_cache[key]
        if name in pkg.native and isinstance(pkg.native[name], RT):
            pkg.rt_cache[key] = pkg.native[name]
            return pkg.native[name]
        if name not in pkg.types:
            raise Unsupported(f"{pkg.path} has no type {name}")
        decl = pkg.types[name]
        ast, imports = decl._file
        rt = RT("named", name=name, pkg=pkg)
        pkg.rt_cache[key] = rt
        cell = {}

        def under():
            if "u" not in cell:
                cell["u"] = self.rtype(decl.type, pkg, imports)
            return cell["u"]
        rt.under = under
        if decl.alias:
            real = under()
            pkg.rt_cache[key] = real
            return real
        return rt

    def zero(self, rt):
        k = rt.kind
        if k == "named":
            u = rt.u()
            if u.kind == "struct":
                return GoStruct(rt, {n: self.zero(ft) for n, ft in u.fields if n is not None} | {self.embedded_name(ft): self.zero(ft) for n, ft in u.fields if n is None})
            return self.zero(u)
        if k == "basic":
            n = rt.name
            if n in NP:
                return NP[n](0)
            if n in PYINT:
                return 0
            if n == "float64":
                return 0.0
            if n == "complex128":
                return 0j
            if n == "bool":
                return False
            if n == "string":
                return ""
            return None
        if k == "struct":
            return GoStruct(rt, {(n if n is not None else self.embedded_name(ft)): self.zero(ft) for n, ft in rt.fields})
        if k == "array":
            return GoArray([self.zero(rt.elem) for _ in range(rt.n)], 0, rt.n, rt.elem)
        return None                                       # ptr, slice, map, func, iface, typeparam

    @staticmethod
    def embedded_name(ft):
        t = ft.elem if ft.kind == "ptr" else ft
        return t.name or getattr(t, "_stub", None) or "embedded"

    def convert(self, rt, v):
        """T(v)."""
        u = rt.u()
        k = u.kind
        if k == "basic":
            n = u.name
            if n in NP:
                if n == "float32":
                    return np.float32(v)
                if isinstance(v, (float, np.floating)):
                    v = int(math.trunc(float(v)))              # Go: float -> integer conversion truncates toward zero
                if isinstance(v, np.generic):
                    return v.astype(NP[n])
                if isinstance(v, bool):
                    raise GoPanic("conversion of bool")
                if isinstance(v, int):
                    bits = 8 * NP[n](0).itemsize
                    m = v & ((1 << bits) - 1)
                    if n.startswith("int") and m >= 1 << (bits - 1):
                        m -= 1 << bits
                    return NP[n](m)
                raise Unsupported(f"convert {type(v).__name__} to {n}")
            if n in PYINT:
                if isinstance(v, (float, np.floating)):
                    if not -9.3e18 < float(v) < 1.9e19:       # Go: the result of an out-of-range float -> integer conversion is implementation-defined
                        raise Unsupported(f"float {v} is outside the range of {n}")
                    r = int(math.trunc(float(v)))
                else:
                    r = int(v)
                if n in ("uint64", "uint", "uintptr"):
                    r &= (1 << 64) - 1
                elif not -(1 << 63) <= r < (1 << 63):
                    r = (r + (1 << 63)) % (1 << 64) - (1 << 63)
                return r
            if n == "float64":
                return float(v)
            if n == "complex128":
                return complex(v)
            if n == "string":
                return v if isinstance(v, str) else chr(int(v))
            if n == "bool":
                return bool(v)
            if n == "unsafe.Pointer":
                return v
        if k == "ptr":
            if isinstance(v, ElemPtr) and u.elem.u().kind == "array":           # (*[N]T)(unsafe.Pointer(&s[i]))
                arr = u.elem.u()
                if v.i + arr.n > len(v.a):
                    raise GoPanic("pointer-to-array view past the end of the backing array")
                return GoPtr(GoArray(v.a, v.i, arr.n, arr.elem))
            return v
        if k == "slice" and isinstance(v, str):
            vals = [np.uint8(b) for b in v.encode("utf-8")]
            return GoSlice(vals, 0, len(vals), len(vals), u.elem)
        if k in ("slice", "struct", "map", "func", "iface", "array", "typeparam"):
            return v
        raise Unsupported(f"conversion to {rt}")

    def coerce(self, rt, v):
        """Assignment of v to a slot of type rt: untyped constants take the slot's type; struct / array VALUES are copied."""
        if rt is None:
            return self.copy_value(v)
        u = rt.u() if rt.kind in ("named",) else rt
        if u.kind == "basic":
            n = u.name
            if n in NP and not isinstance(v, np.generic):
                return self.convert(rt, v)
            if n == "float64" and isinstance(v, int) and not isinstance(v, bool):
                return float(v)
            if n == "complex128" and isinstance(v, (int, float)) and not isinstance(v, bool):
                return complex(v)
            return v
        return self.copy_value(v)

    def copy_value(self, v):
        if isinstance(v, GoStruct):
            return GoStruct(v.t, {k: self.copy_value(x) for k, x in v.f.items()})
        if isinstance(v, GoArray):
            return GoArray([self.copy_value(x) for x in v.a[v.o:v.o + v.n]], 0, v.n, v.et)
        return v

    # ---- the standard-library slice the reference uses
    def _std(self):
        def mk(path, name=None):
            p = Pkg(path, name or path.rsplit("/", 1)[-1])
            p.initialised = True
            self.pkgs[path] = p
            return p
        m = mk("math")
        m.native.update({"Pi": math.pi, "Exp2": Builtin(lambda a: math.pow(2.0, a[0]), "Exp2"), "Round": Builtin(lambda a: go_round(float(a[0])), "Round"),
                         "Mod": Builtin(lambda a: math.fmod(a[0], a[1]), "Mod"), "Floor": Builtin(lambda a: float(math.floor(a[0])), "Floor"),
                         "Ceil": Builtin(lambda a: float(math.ceil(a[0])), "Ceil"), "Sqrt": Builtin(lambda a: math.sqrt(a[0]), "Sqrt"),
                         "Abs": Builtin(lambda a: abs(a[0]), "Abs"), "Pow": Builtin(lambda a: math.pow(a[0], a[1]), "Pow"),
                         "Log2": Builtin(lambda a: math.log2(a[0]), "Log2"), "Log": Builtin(lambda a: math.log(a[0]), "Log"),
                         "Exp": Builtin(lambda a: math.exp(a[0]), "Exp"), "Cos": Builtin(lambda a: math.cos(a[0]), "Cos"),
                         "Sin": Builtin(lambda a: math.sin(a[0]), "Sin"), "Trunc": Builtin(lambda a: float(math.trunc(a[0])), "Trunc"),
                         "MaxUint32": (1 << 32) - 1, "MaxInt32": (1 << 31) - 1, "MaxInt64": (1 << 63) - 1,
                         "Inf": Builtin(lambda a: math.inf if a[0] >= 0 else -math.inf, "Inf")})
        c = mk("math/cmplx", "cmplx")
        c.native.update({"Exp": Builtin(lambda a: cmath.exp(a[0]), "Exp"), "Abs": Builtin(lambda a: abs(a[0]), "Abs"),
                         "Conj": Builtin(lambda a: a[0].conjugate(), "Conj")})
        un = mk("unsafe")
        un.native["Pointer"] = RT_UPTR
        s = mk("sync")
        s.native["WaitGroup"] = RT("struct", fields=[])
        s.native["Mutex"] = RT("struct", fields=[])
        s.native["RWMutex"] = RT("struct", fields=[])
        s.native["Pool"] = RT("struct", fields=[("New", RT("func"))])
        f = mk("fmt")
        # fmt.Print*: dropped unless self.stdout is a list (then one entry per call: what a reference PROGRAM prints is its result)
        self.stdout = None
        gostr = lambda v: ("true" if v else "false") if isinstance(v, (bool, np.bool_)) else str(v)

        def _println(a):
            if self.stdout is not None:
                self.stdout.append(" ".join(gostr(x) for x in a))

        def _printf(a):
            if self.stdout is not None:
                self.stdout.append(self.sprintf(a))
        f.native.update({"Sprintf": Builtin(lambda a: self.sprintf(a), "Sprintf"), "Println": Builtin(_println, "Println"),
                         "Printf": Builtin(_printf, "Printf"), "Print": Builtin(_println, "Print")})
        r = mk("math/rand", "rand")
        RAND = RT("struct", fields=[])
        r.native["Rand"] = RAND
        r.native.update({"Int63": Builtin(lambda a: int(self.rng.randint(0, 1 << 62)), "Int63"),
                         "NewSource": Builtin(lambda a: ("source", a[0]), "NewSource"),
                         "New": Builtin(lambda a: GoPtr(GoStruct(RAND, {})), "New"),
                         "Seed": Builtin(lambda a: None, "Seed"),
                         "Uint32": Builtin(lambda a: np.uint32(self.rng.randint(0, 1 << 32, dtype=np.uint64)), "Uint32"),
                         "Intn": Builtin(lambda a: int(self.rng.randint(0, int(a[0]))), "Intn"),
                         "Float64": Builtin(lambda a: float(self.rng.random_sample()), "Float64"),
                         "NormFloat64": Builtin(lambda a: float(self.rng.standard_normal()), "NormFloat64")})
        self.RAND = RAND
        t = mk("time")
        t.native.update({"Now": Builtin(lambda a: 0, "Now"), "Since": Builtin(lambda a: 0, "Since"), "Duration": BASIC_RT["int64"]})
        rt_ = mk("runtime")
        rt_.native.update({"NumCPU": Builtin(lambda a: 1, "NumCPU"), "GOMAXPROCS": Builtin(lambda a: 1, "GOMAXPROCS"),
                           "LockOSThread": Builtin(lambda a: None, "LockOSThread"), "UnlockOSThread": Builtin(lambda a: None, "UnlockOSThread")})
        # ---- what tools/go_golden/main.go needs to RUN under the interpreter (files really get written: numpy reads them back)
        import struct as _struct
        FILE = RT("struct", fields=[])
        FILE._stub = "File"
        self.FILE_T = FILE
        osp = mk("os")
        osp.native["File"] = FILE

        def os_create(a):
            return (GoPtr(GoStruct(FILE, {"$fh": open(a[0], "wb")})), None)
        osp.native.update({"Create": Builtin(os_create, "Create"), "MkdirAll": Builtin(lambda a: os.makedirs(a[0], exist_ok=True), "MkdirAll")})
        fp = mk("path/filepath", "filepath")
        fp.native["Join"] = Builtin(lambda a: os.path.join(*a), "Join")
        st = mk("strings")
        st.native.update({"Join": Builtin(lambda a: a[1].join(a[0].a[a[0].o:a[0].o + a[0].n]), "Join"), "Repeat": Builtin(lambda a: a[0] * int(a[1]), "Repeat")})
        m.native["Float64bits"] = Builtin(lambda a: _struct.unpack("<Q", _struct.pack("<d", float(a[0])))[0], "Float64bits")
        ORDER = RT("struct", fields=[])
        ORDER._stub = "littleEndian"
        self.ORDER_T = ORDER
        bn = mk("encoding/binary", "binary")
        bn.native["LittleEndian"] = GoStruct(ORDER, {})

        def bin_write(a):
            v = a[2]
            if isinstance(v, np.generic):
                gi_bytes = v.tobytes()
            else:
                raise Unsupported("binary.Write of this value")
            a[0].v.f["$fh"].write(gi_bytes)
            return None
        bn.native["Write"] = Builtin(bin_write, "Write")
        fl = mk("flag")
        self.flag_overrides = {}

        def flag_var(a):
            box = Env()
            box.vars["v"] = self.flag_overrides.get(a[0], a[1])
            return VarPtr(box, "v")
        for nme in ("String", "Int", "Int64", "Bool", "Float64"):
            fl.native[nme] = Builtin(flag_var, nme)
        fl.native["Parse"] = Builtin(lambda a: None, "Parse")
        tst = mk("testing")
        self.TEST_T = RT("struct", fields=[("name", BASIC_RT["string"])])
        self.TEST_T._stub = "T"
        tst.native["T"] = self.TEST_T
        tst.native["B"] = RT("struct", fields=[])
        tst.native["Short"] = Builtin(lambda a: False, "Short")
        at = mk("sync/atomic", "atomic")

        def add_u32(a):
            v = np.uint32(int(ptr_load(a[0])) + int(a[1]))
            ptr_store(a[0], v)
            return v
        at.native["AddUint32"] = Builtin(add_u32, "AddUint32")
        for nme in ("WaitGroup", "Mutex", "RWMutex", "Pool"):
            s.native[nme]._stub = nme

    @staticmethod
    def sprintf(a):
        import re as _re
        try:
            args = list(a[1:])

            def sub(m):
                if m.group(0) == "%%":
                    return "%"
                v = args.pop(0) if args else "%!MISSING"
                verb = m.group(0)[-1]
                if verb in "xX" and not isinstance(v, (str, float)):
                    return format(int(v), m.group(0)[1:-1] + verb)
                if verb in "feg" and isinstance(v, (int, float, np.generic)):
                    return format(float(v), m.group(0)[1:-1] + verb if m.group(0)[1:-1] else ".6f")
                if verb in "db" and isinstance(v, (int, np.integer)) and not isinstance(v, (bool, np.bool_)):
                    return format(int(v), m.group(0)[1:-1] + verb)
                if isinstance(v, (bool, np.bool_)):
                    return "true" if v else "false"
                if verb == "s" and m.group(0)[1:-1]:
                    return format(str(v), m.group(0)[1:-1].replace("-", "<") if "-" in m.group(0) else ">" + m.group(0)[1:-1])
                return str(v)
            return _re.sub(r"%[-+0-9. #]*[vdsfxXtqegb%T]", sub, a[0])
        except Exception:                               # noqa: BLE001
            return str(a)

    # native methods of the stub types
    def native_method(self, recv, name):
        t = recv.t if isinstance(recv, GoStruct) else None
        if t is getattr(self, "FILE_T", None):
            fh = recv.f["$fh"]

            def fwrite(a):
                v = a[0]
                data = v.encode("latin-1") if isinstance(v, str) else bytes(int(x) & 0xFF for x in v.a[v.o:v.o + v.n])
                fh.write(data)
                return (len(data), None)
            return {"Write": fwrite, "Close": lambda a: fh.close()}.get(name)
        if t is getattr(self, "ORDER_T", None):
            def put(nbytes):
                def f(a):
                    b, v = a[0], int(a[1])
                    for k in range(nbytes):
                        b.a[b.o + k] = np.uint8((v >> (8 * k)) & 0xFF)
                return f
            return {"PutUint32": put(4), "PutUint64": put(8), "PutUint16": put(2)}.get(name)
        if t is self.RAND:
            return {"Uint32": lambda a: np.uint32(self.rng.randint(0, 1 << 32, dtype=np.uint64)),
                    "Intn": lambda a: int(self.rng.randint(0, int(a[0]))),
                    "Int63": lambda a: int(self.rng.randint(0, 1 << 62)),
                    "Float64": lambda a: float(self.rng.random_sample()),
                    "NormFloat64": lambda a: float(self.rng.standard_normal())}.get(name)
        if t is getattr(self, "TEST_T", None):
            log = recv.f.setdefault("$log", {"failures": [], "logs": []})

            def fail(a, fatal=False):
                log["failures"].append(self.sprintf(a) if a and isinstance(a[0], str) and "%" in a[0] else " ".join(str(x) for x in a))
                if fatal:
                    raise TestFatal()

            def run_sub(a):
                sub = GoPtr(GoStruct(self.TEST_T, {"name": recv.f["name"] + "/" + str(a[0]), "$log": log}))
                try:
                    self.call(a[1], [sub])
                except TestFatal:
                    pass
                return True

            def skip(a):
                log["logs"].append("SKIP " + " ".join(str(x) for x in a))
                raise TestFatal(skipped=True)
            return {"Errorf": lambda a: fail(a), "Error": lambda a: fail(a), "Fatalf": lambda a: fail(a, True), "Fatal": lambda a: fail(a, True),
                    "Fail": lambda a: fail(["Fail()"]), "FailNow": lambda a: fail(["FailNow()"], True),
                    "Logf": lambda a: log["logs"].append(self.sprintf(a)), "Log": lambda a: log["logs"].append(" ".join(str(x) for x in a)),
                    "Run": run_sub, "Skip": skip, "Skipf": skip, "SkipNow": skip, "Helper": lambda a: None, "Name": lambda a: recv.f["name"],
                    "Parallel": lambda a: None, "Cleanup": lambda a: None}.get(name)
        sync = self.pkgs["sync"]
        if t is sync.native["WaitGroup"]:
            return (lambda a: None) if name in ("Add", "Done", "Wait") else None
        if t in (sync.native["Mutex"], sync.native["RWMutex"]):
            return (lambda a: None) if name in ("Lock", "Unlock", "RLock", "RUnlock") else None
        if t is sync.native["Pool"]:
            if name == "Get":
                return lambda a: self.call(recv.f["New"], [])
            if name == "Put":
                return lambda a: None
        return None

    # ---- calling
    def call_func(self, pkg_name, func, *args):
        """Entry point for drivers: call a package-level function of a reference package by name."""
        pkg = self.pkgs.get(f"{self.module}/{pkg_name}") or self.load(pkg_name)
        self.ensure_init(pkg)
        return self.call_decl(pkg.funcs[func], pkg, list(args), None)

    def run_reference_tests(self, pkg_name, only=None, pkg=None, directory=None):
        """Runs the reference's own Test* functions of one package (its *_test.go files: `package x` tests join the package, `package
        x_test` tests become a package of their own that imports it) with a testing.T stand-in.  Returns {test name: {"failures": [...],
        "skipped": bool, "statements": n}}; a test that panics is reported as a failure with the panic's message."""
        # (pkg, directory: a package that does not live under the reference root -- the shim's own Go test, shim/go/gates)
        pkg = pkg or self.load(pkg_name)
        d = directory or os.path.join(self.root, pkg_name)
        ext = None
        loaded = getattr(pkg, "_tests_loaded", False)
        if not loaded:
            pkg._tests_loaded = True
            for f in sorted(os.listdir(d)):
                if not f.endswith("_test.go"):
                    continue
                ast = gocheck.parse_source(open(os.path.join(d, f)).read(), os.path.join(d, f), bodies=False)
                if ast.package == pkg_name or ast.package == pkg.name:
                    self._add_file(pkg, ast)
                else:
                    ext = self.pkgs.get(pkg.path + "_test") or Pkg(pkg.path + "_test", ast.package)
                    self.pkgs[pkg.path + "_test"] = ext
                    self._add_file(ext, ast)
        ext = self.pkgs.get(pkg.path + "_test")
        out = {}
        for host in [p for p in (pkg, ext) if p is not None]:
            self.ensure_init(host)
            for name, decl in sorted(host.funcs.items()):
                if not name.startswith("Test") or len(decl.sig.params) != 1 or (only and name not in only):
                    continue
                if not getattr(decl._file[0], "fname", "").endswith("_test.go"):
                    continue
                t = GoPtr(GoStruct(self.TEST_T, {"name": name}))
                s0 = self.steps
                skipped = False
                try:
                    self.call_decl(decl, host, [t], None)
                except TestFatal as e:
                    skipped = e.skipped
                except GoPanic as e:
                    t.v.f.setdefault("$log", {"failures": [], "logs": []})["failures"].append(f"panic: {e}")
                log = t.v.f.get("$log", {"failures": [], "logs": []})
                out[name] = {"failures": list(log["failures"]), "skipped": skipped, "statements": self.steps - s0}
        return out

    def call_method(self, recv, name, *args):
        fn = self.member(recv, name, None)
        return self.call(fn, list(args))

    def call(self, fn, args):
        if isinstance(fn, Builtin):
            return fn.fn(args)
        if isinstance(fn, GoFunc):
            return self.call_decl(fn.decl, fn.pkg, args, fn.recv, fn.env)
        if callable(fn):
            return fn(args)
        raise GoPanic(f"call of a non-function {fn!r}")

    def call_decl(self, decl, pkg, args, recv, closure_env=None):
        if decl.kind == "funcdecl":
            ast, imports = decl._file
            if decl.body is None:
                if decl.lazy is None:
                    raise Unsupported(f"{decl.name} has no body")
                parser, pos = decl.lazy
                decl.body = parser.parse_body_at(pos)
            sig, body = decl.sig, decl.body
            env = Env(pkg.env)
        else:                                             # funclit
            ast, imports = decl._file
            sig, body = decl.sig, decl.body
            env = Env(closure_env)
        env.vars["$file"] = (pkg, imports)
        if decl.kind == "funcdecl" and decl.recv is not None:
            rname = decl.recv.name
            if rname:
                if decl.recv.type.kind == "tptr":
                    env.vars[rname] = recv if isinstance(recv, GoPtr) else GoPtr(recv)
                else:
                    base = recv.v if isinstance(recv, GoPtr) else recv
                    env.vars[rname] = self.copy_value(base)
        params = sig.params
        if params and params[-1].variadic:
            fixed = len(params) - 1
            rest = args[fixed:]
            if len(rest) == 1 and isinstance(rest[0], tuple) and rest[0] and rest[0][0] == "$spread":
                var = rest[0][1]
            else:
                et = self.rtype(params[-1].type, pkg, imports)
                var = GoSlice([self.coerce(et, x) for x in rest], 0, len(rest), len(rest), et)
            args = args[:fixed] + [var]
        if len(args) != len(params):
            raise GoPanic(f"{getattr(decl, 'name', 'func literal')}: {len(args)} arguments for {len(params)} parameters")
        for p, a in zip(params, args):
            if p.name and p.name != "_":
                if p.variadic:
                    env.vars[p.name] = a
                else:
                    env.vars[p.name] = self.coerce(self.ptype(p, pkg, imports), a)
        frame = Frame()
        named = [r for r in sig.results if r.name]
        if named:
            frame.named = [r.name for r in sig.results]
            for r in sig.results:
                env.vars[r.name] = self.zero(self.rtype(r.type, pkg, imports))
        env.vars["$frame"] = frame
        try:
            sig_ = self.exec_block(body, env, new_scope=False)
        finally:
            while frame.defers:
                fn, a = frame.defers.pop()
                self.call(fn, a)
        if sig_ == RETURN:
            res = frame.results
        else:
            res = ()
        if frame.named and not res:
            res = tuple(env.vars[n] for n in frame.named)
        if len(sig.results) == 0:
            return None
        if len(sig.results) == 1:
            rtt = self.ptype(sig.results[0], pkg, imports)
            return self.coerce(rtt, res[0])
        return tuple(self.coerce(self.ptype(r, pkg, imports), v) for r, v in zip(sig.results, res))

    def ptype(self, p, pkg, imports):
        c = getattr(p, "_rt", None)
        if c is None:
            c = p._rt = self.rtype(p.type, pkg, imports)
        return c

    # ---- statements
    def exec_block(self, block, env, new_scope=True):
        e = Env(env) if new_scope else env
        for st in block.stmts:
            s = self.exec(st, e)
            if s is not None:
                return s
        return None

    def exec(self, st, env):
        self.steps += 1
        k = st.kind
        if k == "exprstmt":
            self.eval(st.x, env)
            return None
        if k == "assign":
            return self.exec_assign(st, env)
        if k == "incdec":
            cur = self.eval(st.x, env)
            self.store(st.x, self.binop("+" if st.op == "++" else "-", cur, 1), env)
            return None
        if k == "return":
            vals = [self.eval(v, env) for v in st.values]
            if len(vals) == 1 and isinstance(vals[0], tuple) and not (vals[0] and vals[0][0] == "$spread"):
                vals = list(vals[0])
            self.frame(env).results = tuple(vals)
            return RETURN
        if k == "if":
            e = Env(env)
            if st.init is not None:
                self.exec(st.init, e)
            if self.eval(st.cond, e):
                return self.exec_block(st.then, e)
            if st.els is not None:
                return self.exec(st.els, e) if st.els.kind == "if" else self.exec_block(st.els, e)
            return None
        if k == "block":
            return self.exec_block(st, env)
        if k == "for":
            e = Env(env)
            if st.init is not None:
                self.exec(st.init, e)
            while st.cond is None or self.eval(st.cond, e):
                s = self.exec_block(st.body, e)
                if s == BREAK:
                    break
                if s == RETURN:
                    return s
                if st.post is not None:
                    self.exec(st.post, e)
            return None
        if k == "forrange":
            return self.exec_range(st, env)
        if k == "declstmt":
            for sp in st.specs:
                pkg, imports = self.file_of(env)
                vals = [self.eval(v, env) for v in sp.values]
                if len(vals) == 1 and isinstance(vals[0], tuple) and len(sp.names) > 1:
                    vals = list(vals[0])
                rt = self.rtype(sp.type, pkg, imports) if sp.type is not None else None
                for i, n in enumerate(sp.names):
                    if n == "_":
                        continue
                    if vals:
                        env.vars[n] = self.coerce(rt, vals[i]) if rt is not None else self.copy_value(vals[i])
                    else:
                        env.vars[n] = self.zero(rt)
            return None
        if k == "branch":
            return BREAK if st.what == "break" else CONTINUE
        if k == "defer":
            c = st.call
            fn = self.eval_callee(c.fun, env)
            args = [self.eval(a, env) for a in c.args]
            self.frame(env).defers.append((fn, args))
            return None
        if k == "go":
            self.eval(st.call, env)                       # goroutines run inline: deterministic
            return None
        if k == "switch":
            e = Env(env)
            if st.init is not None:
                self.exec(st.init, e)
            tag = self.eval(st.tag, e) if st.tag is not None else True
            default = None
            for exprs, body in st.clauses:
                if exprs is None:
                    default = body
                    continue
                for x in exprs:
                    v = self.eval(x, e)
                    if (v == tag) if st.tag is not None else bool(v):
                        s = self.exec_block(body, e)
                        return None if s == BREAK else s
            if default is not None:
                s = self.exec_block(default, e)
                return None if s == BREAK else s
            return None
        raise Unsupported(f"statement {k} (line {st.line})")

    def frame(self, env):
        e = env.find("$frame")
        return e.vars["$frame"]

    def file_of(self, env):
        e = env.find("$file")
        return e.vars["$file"]

    def exec_range(self, st, env):
        x = self.eval(st.x, env)
        e = Env(env)
        if isinstance(x, GoPtr) and isinstance(x.v, GoArray):
            x = x.v
        if x is None:
            items = []
        elif isinstance(x, (GoSlice, GoArray)):
            items = None
        elif isinstance(x, dict):
            items = list(x.items())
        elif isinstance(x, str):
            items = list(enumerate(x))
        else:
            raise Unsupported(f"range over {type(x).__name__}")
        n = x.n if items is None else len(items)
        for idx in range(n):
            if items is None:
                kv = (idx, x.a[x.o + idx]) if len(st.lhs) > 1 else (idx, None)
            else:
                kv = items[idx]
            for lhs, v in zip(st.lhs, kv):
                if lhs.kind == "ident" and lhs.name == "_":
                    continue
                if st.define:
                    e.vars[lhs.name] = self.copy_value(v)
                else:
                    self.store(lhs, v, e)
            s = self.exec_block(st.body, e)
            if s == BREAK:
                break
            if s == RETURN:
                return s
        return None

    def exec_assign(self, st, env):
        op = st.op
        if op in ("=", ":="):
            if len(st.rhs) == 1 and len(st.lhs) > 1:
                r = st.rhs[0]
                if r.kind == "index":                      # v, ok := m[k]
                    m = self.eval(r.x, env)
                    if isinstance(m, dict):
                        kx = self.eval(r.index, env)
                        vals = [m.get(kx), kx in m]
                    else:
                        vals = list(self.eval(r, env))
                elif r.kind == "typeassert":
                    v = self.eval(r.x, env)
                    vals = [v, v is not None]
                else:
                    vals = list(self.eval(r, env))
            else:
                vals = [self.eval(r, env) for r in st.rhs]
            if len(vals) != len(st.lhs):
                raise GoPanic(f"assignment mismatch at line {st.line}")
            if len(vals) > 1:
                vals = [self.copy_value(v) for v in vals]
            for lhs, v in zip(st.lhs, vals):
                if lhs.kind == "ident" and lhs.name == "_":
                    continue
                if op == ":=" and lhs.kind == "ident" and lhs.name not in env.vars:
                    env.vars[lhs.name] = self.copy_value(v)
                else:
                    self.store(lhs, v, env)
            return None
        cur = self.eval(st.lhs[0], env)
        v = self.eval(st.rhs[0], env)
        self.store(st.lhs[0], self.binop(op[:-1], cur, v), env)
        return None

    def store(self, lhs, v, env):
        k = lhs.kind
        if k == "ident":
            e = env.find(lhs.name)
            if e is None:
                pkg, _ = self.file_of(env)
                if lhs.name in pkg.value_decls or lhs.name in pkg.values:
                    old = self.pkg_value(pkg, lhs.name)
                    pkg.values[lhs.name] = self.like(old, v)
                    return
                raise GoPanic(f"assignment to undeclared {lhs.name}")
            e.vars[lhs.name] = self.like(e.vars[lhs.name], v)
            return
        if k == "index":
            base = self.eval(lhs.x, env)
            i = self.eval(lhs.index, env)
            if isinstance(base, GoPtr):
                base = base.v
            if isinstance(base, dict):
                base[i] = self.copy_value(v)
                return
            i = int(i)
            if not 0 <= i < base.n:
                raise GoPanic(f"index out of range [{i}] with length {base.n} (line {lhs.line})")
            base.a[base.o + i] = self.coerce(base.et, v)
            return
        if k == "selector":
            target = self.eval(lhs.x, env)
            if isinstance(target, PkgRef):
                old = self.pkg_value(target.pkg, lhs.sel)
                target.pkg.values[lhs.sel] = self.like(old, v)
                return
            if isinstance(target, GoPtr):
                target = target.v
            if not isinstance(target, GoStruct) or lhs.sel not in target.f:
                raise GoPanic(f"no field {lhs.sel} to assign (line {lhs.line})")
            target.f[lhs.sel] = self.like(target.f[lhs.sel], v, self.field_type(target, lhs.sel))
            return
        if k == "unary" and lhs.op == "*":
            p = self.eval(lhs.x, env)
            if isinstance(p, GoPtr):
                src = v
                if isinstance(p.v, GoStruct) and isinstance(src, GoStruct):
                    p.v.f = self.copy_value(src).f
                    return
                raise Unsupported("store through this pointer")
            if isinstance(p, ElemPtr):
                p.a[p.i] = self.coerce(p.et, v)
                return
            if isinstance(p, VarPtr):
                p.env.vars[p.name] = self.like(p.env.vars[p.name], v)
                return
            if isinstance(p, FieldPtr):
                p.s.f[p.name] = self.like(p.s.f[p.name], v, self.field_type(p.s, p.name))
                return
        if k == "paren":
            return self.store(lhs.x, v, env)
        raise Unsupported(f"assignment target {k}")

    def field_type(self, st, name):
        u = st.t.u()
        for n, ft in u.fields:
            if n == name:
                return ft
        return None

    def like(self, old, v, rt=None):
        """Keep the slot's numeric type when an untyped constant (Python number) is stored into a typed slot."""
        if rt is not None:
            return self.coerce(rt, v)
        if isinstance(old, np.generic) and not isinstance(v, np.generic):
            if isinstance(v, (int, float)) and not isinstance(v, bool):
                return type(old)(v) if isinstance(old, np.floating) else self.convert(BASIC_RT[old.dtype.name], v)
        if isinstance(old, float) and isinstance(v, int) and not isinstance(v, bool):
            return float(v)
        if isinstance(old, complex) and isinstance(v, (int, float)) and not isinstance(v, bool):
            return complex(v)
        return self.copy_value(v)

    # ---- expressions
    def eval(self, e, env):
        k = e.kind
        if k == "ident":
            return self.ident(e.name, env, e)
        if k == "lit":
            c = getattr(e, "_c", None)
            if c is None:
                t = e.text
                if e.lkind == "int":
                    c = int(t.replace("_", ""), 0)
                elif e.lkind == "float":
                    c = float(t.replace("_", ""))
                elif e.lkind == "string":
                    c = t[1:-1] if t[0] == "`" or "\\" not in t else t[1:-1].encode("latin-1", "backslashreplace").decode("unicode_escape")
                else:
                    c = ord(t[1:-1].encode("latin-1", "backslashreplace").decode("unicode_escape"))
                e._c = (c,)
                return c
            return c[0]
        if k == "binary":
            op = e.op
            if op == "&&":
                return bool(self.eval(e.x, env)) and bool(self.eval(e.y, env))
            if op == "||":
                return bool(self.eval(e.x, env)) or bool(self.eval(e.y, env))
            return self.binop(op, self.eval(e.x, env), self.eval(e.y, env))
        if k == "index":
            base = self.eval(e.x, env)
            i = self.eval(e.index, env)
            if isinstance(base, GoPtr):
                base = base.v
            if isinstance(base, dict):
                if i in base:
                    return base[i]
                raise Unsupported("zero value of a missing map key")
            if isinstance(base, str):
                return np.uint8(ord(base[int(i)]))
            i = int(i)
            if base is None or not 0 <= i < base.n:
                raise GoPanic(f"index out of range [{i}] with length {0 if base is None else base.n} (line {e.line})")
            return base.a[base.o + i]
        if k == "selector":
            return self.selector(e, env)
        if k == "call":
            return self.call_expr(e, env)
        if k == "paren":
            return self.eval(e.x, env)
        if k == "unary":
            return self.unary(e, env)
        if k == "slice":
            return self.slice_expr(e, env)
        if k == "complit":
            pkg, imports = self.file_of(env)
            return self.complit(e, env, self.type_of_expr(e.type, env))
        if k == "funclit":
            e._file = (self.file_of(env)[0].files[0][0], self.file_of(env)[1])
            return GoFunc(e, env, self.file_of(env)[0])
        if k == "typeassert":
            return self.eval(e.x, env)
        if k in ("tslice", "tarray", "tmap", "tstruct", "tptr", "tname", "functype", "tiface"):
            pkg, imports = self.file_of(env)
            return TypeVal(self.rtype(e, pkg, imports))
        raise Unsupported(f"expression {k} (line {e.line})")

    def ident(self, name, env, node=None):
        e = env.find(name)
        if e is not None:
            return e.vars[name]
        pkg, imports = self.file_of(env)
        if name in pkg.value_decls or name in pkg.values:
            return self.pkg_value(pkg, name)
        if name in pkg.funcs:
            return GoFunc(pkg.funcs[name], None, pkg, name=name)
        if name in pkg.types:
            return TypeVal(self.named(pkg, name))
        if name in imports:
            return PkgRef(self.pkg_by_import(imports[name]))
        if name in BASIC_RT:
            return TypeVal(BASIC_RT[name])
        if name == "true":
            return True
        if name == "false":
            return False
        if name == "nil":
            return None
        if name in BUILTINS:
            return Builtin(None, name)
        raise GoPanic(f"undefined: {name}")

    def pkg_value(self, pkg, name):
        if name in pkg.values:
            return pkg.values[name]
        if name in pkg.native:
            return pkg.native[name]
        if name not in pkg.value_decls:
            raise GoPanic(f"{pkg.name}.{name} undefined")
        spec, i = pkg.value_decls[name]
        ast, imports = spec._file
        env = Env(pkg.env)
        env.vars["$file"] = (pkg, imports)
        env.vars["$frame"] = Frame()
        rt = self.rtype(spec.type, pkg, imports) if spec.type is not None else None
        if spec.values:
            if len(spec.values) == len(spec.names):
                v = self.eval(spec.values[i], env)
            else:
                v = self.eval(spec.values[0], env)[i]
            v = self.coerce(rt, v) if rt is not None else self.copy_value(v)
        else:
            v = self.zero(rt)
        pkg.values[name] = v
        return v

    def selector(self, e, env):
        x = self.eval(e.x, env)
        if isinstance(x, PkgRef):
            p = x.pkg
            self.ensure_init(p)
            n = e.sel
            if n in p.native:
                v = p.native[n]
                return TypeVal(v) if isinstance(v, RT) else v
            if n in p.value_decls or n in p.values:
                return self.pkg_value(p, n)
            if n in p.funcs:
                return GoFunc(p.funcs[n], None, p, name=n)
            if n in p.types:
                return TypeVal(self.named(p, n))
            raise GoPanic(f"{p.name}.{n} undefined")
        return self.member(x, e.sel, e)

    def member(self, x, name, node):
        target = x.v if isinstance(x, GoPtr) else x
        if isinstance(target, GoStruct):
            if name in target.f:
                return target.f[name]
            nm = self.native_method(target, name)
            if nm is not None:
                return Builtin(nm, name)
            t = target.t
            if t.kind == "named":
                decl = t.pkg.methods.get((t.name, name))
                if decl is not None:
                    return GoFunc(decl, None, t.pkg, recv=x if isinstance(x, GoPtr) else target, name=name)
            for fn, fv in target.f.items():                      # promoted through an embedded struct
                inner = fv.v if isinstance(fv, GoPtr) else fv
                if isinstance(inner, GoStruct):
                    nm = self.native_method(inner, name)
                    if nm is not None:
                        return Builtin(nm, name)
                if isinstance(inner, GoStruct) and inner.t.kind == "named" and inner.t.name == fn:
                    try:
                        return self.member(fv, name, node)
                    except GoPanic:
                        pass
        raise GoPanic(f"no field or method {name} on {type(target).__name__} (line {getattr(node, 'line', '?')})")

    def unary(self, e, env):
        op = e.op
        if op == "&":
            t = e.x
            while t.kind == "paren":
                t = t.x
            if t.kind == "complit":
                return GoPtr(self.eval(t, env))
            if t.kind == "index":
                base = self.eval(t.x, env)
                i = int(self.eval(t.index, env))
                if isinstance(base, GoPtr):
                    base = base.v
                el = base.a[base.o + i] if 0 <= i < base.n else None
                if isinstance(el, (GoStruct, GoArray)):
                    return GoPtr(el)
                if not 0 <= i < base.n:
                    raise GoPanic(f"index out of range [{i}] with length {base.n} (line {e.line})")
                return ElemPtr(base.a, base.o + i, base.et)
            v = self.eval(t, env)
            if isinstance(v, (GoStruct, GoArray)):
                return GoPtr(v)
            if t.kind == "ident":
                return VarPtr(env.find(t.name), t.name)
            if t.kind == "selector":
                owner = self.eval(t.x, env)
                owner = owner.v if isinstance(owner, GoPtr) else owner
                if isinstance(owner, GoStruct) and t.sel in owner.f:
                    return FieldPtr(owner, t.sel)
            raise Unsupported(f"address of this expression (line {e.line})")
        if op == "*":
            v = self.eval(e.x, env)
            if isinstance(v, TypeVal):
                return TypeVal(RT("ptr", elem=v.rt))
            if isinstance(v, GoPtr):
                return v.v
            if isinstance(v, ElemPtr):
                return v.a[v.i]
            if isinstance(v, VarPtr):
                return v.env.vars[v.name]
            if isinstance(v, FieldPtr):
                return v.s.f[v.name]
            raise GoPanic(f"nil pointer dereference (line {e.line})")
        v = self.eval(e.x, env)
        if op == "-":
            return -v
        if op == "+":
            return v
        if op == "!":
            return not v
        if op == "^":
            return ~v
        raise Unsupported(f"unary {op}")

    def binop(self, op, a, b):
        an, bn = isinstance(a, np.generic), isinstance(b, np.generic)
        if op in ("<<", ">>"):
            sh = int(b)
            if an:
                if sh >= 8 * a.itemsize:
                    return type(a)(0) if op == "<<" or a >= 0 else type(a)(-1)
                return type(a)(a << type(a)(sh)) if op == "<<" else type(a)(a >> type(a)(sh))
            r = a << sh if op == "<<" else a >> sh
            if type(r) is int and not -9223372036854775808 <= r <= 18446744073709551615:
                raise Unsupported(f"64-bit integer overflow in {a} << {sh}: Go would wrap here, the interpreter does not model it")
            return r
        if an and not bn and isinstance(b, int) and not isinstance(b, bool) and isinstance(a, np.integer):
            b = self.convert(BASIC_RT[a.dtype.name], b)
        elif bn and not an and isinstance(a, int) and not isinstance(a, bool) and isinstance(b, np.integer):
            a = self.convert(BASIC_RT[b.dtype.name], a)
        if op in ("+", "-", "*"):
            r = a + b if op == "+" else a - b if op == "-" else a * b
            # Go's int / int64 / uint64 are Python ints here: a result outside 64 bits would WRAP in Go and not here -- never silently
            if type(r) is int and not -9223372036854775808 <= r <= 18446744073709551615:
                raise Unsupported(f"64-bit integer overflow in {a} {op} {b}: Go would wrap here, the interpreter does not model it")
            return r
        if op == "/":
            if isinstance(a, (float, complex)) or isinstance(b, (float, complex)) or isinstance(a, np.floating):
                return a / b
            if an or bn:
                if b == 0:
                    raise GoPanic("integer divide by zero")
                if isinstance(a, np.unsignedinteger):
                    return a // b
                return type(a)(trunc_div(int(a), int(b)))
            return trunc_div(a, b)
        if op == "%":
            if an or bn:
                if isinstance(a, np.unsignedinteger):
                    return a % b
                return type(a)(int(a) - trunc_div(int(a), int(b)) * int(b))
            return a - trunc_div(a, b) * b
        if op == "==":
            return self.equal(a, b)
        if op == "!=":
            return not self.equal(a, b)
        if op == "<":
            return bool(a < b)
        if op == "<=":
            return bool(a <= b)
        if op == ">":
            return bool(a > b)
        if op == ">=":
            return bool(a >= b)
        if op == "&":
            return a & b
        if op == "|":
            return a | b
        if op == "^":
            return a ^ b
        if op == "&^":
            return a & ~b
        raise Unsupported(f"operator {op}")

    @staticmethod
    def equal(a, b):
        if a is None or b is None:
            return a is b
        if isinstance(a, (GoPtr, GoSlice, GoFunc)) or isinstance(b, (GoPtr, GoSlice, GoFunc)):
            if isinstance(a, GoPtr) and isinstance(b, GoPtr):
                return a.v is b.v
            return a is b
        if isinstance(a, GoStruct) and isinstance(b, GoStruct):
            return a.f == b.f
        return bool(a == b)

    def slice_expr(self, e, env):
        base = self.eval(e.x, env)
        if isinstance(base, GoPtr):
            base = base.v
        parts = [None if p is None else int(self.eval(p, env)) for p in e.parts]
        if isinstance(base, str):
            return base[parts[0] or 0:parts[1] if parts[1] is not None else len(base)]
        if base is None:
            if any(p for p in parts):
                raise GoPanic("slice of nil")
            return None
        cap = base.c if isinstance(base, GoSlice) else base.n
        lo = parts[0] or 0
        hi = parts[1] if len(parts) > 1 and parts[1] is not None else base.n
        mx = parts[2] if len(parts) > 2 and parts[2] is not None else cap
        if not 0 <= lo <= hi <= mx <= cap:
            raise GoPanic(f"slice bounds out of range [{lo}:{hi}:{mx}] with capacity {cap} (line {e.line})")
        return GoSlice(base.a, base.o + lo, hi - lo, mx - lo, base.et)

    def type_of_expr(self, tnode, env):
        if tnode is None:
            return None
        v = self.eval(tnode, env)
        if not isinstance(v, TypeVal):
            raise GoPanic(f"not a type in a composite literal (line {tnode.line})")
        return v.rt

    def complit(self, e, env, rt):
        u = rt.u()
        if u.kind == "ptr":                                  # elided &T{...} inside []*T{{...}}
            return GoPtr(self.complit(e, env, u.elem))
        if u.kind == "struct":
            st = self.zero(rt)
            if not isinstance(st, GoStruct):
                st = GoStruct(rt, {})
            ftypes = {n if n is not None else self.embedded_name(ft): ft for n, ft in u.fields}
            order = [n if n is not None else self.embedded_name(ft) for n, ft in u.fields]
            for i, (k, v) in enumerate(e.elts):
                name = k.name if k is not None else order[i]
                ft = ftypes[name]
                val = self.complit(v, env, ft) if (v.kind == "complit" and v.type is None) else self.eval(v, env)
                st.f[name] = self.coerce(ft, val)
            return st
        if u.kind in ("slice", "array"):
            et = u.elem
            vals = []
            for k, v in e.elts:
                if k is not None:
                    raise Unsupported("indexed slice literal")
                val = self.complit(v, env, et) if (v.kind == "complit" and v.type is None) else self.eval(v, env)
                vals.append(self.coerce(et, val))
            if u.kind == "array":
                vals += [self.zero(et) for _ in range(u.n - len(vals))]
                return GoArray(vals, 0, u.n, et)
            return GoSlice(vals, 0, len(vals), len(vals), et)
        if u.kind == "map":
            out = {}
            for k, v in e.elts:
                kk = self.eval(k, env)
                out[kk] = self.coerce(u.elem, self.complit(v, env, u.elem) if (v.kind == "complit" and v.type is None) else self.eval(v, env))
            return out
        raise Unsupported(f"composite literal of {rt}")

    def eval_callee(self, fun, env):
        return self.eval(fun, env)

    def call_expr(self, e, env):
        f = e.fun
        if f.kind == "ident" and f.name in BUILTINS and env.find(f.name) is None:
            return self.builtin(f.name, e, env)
        fn = self.eval(f, env)
        if isinstance(fn, TypeVal):
            if len(e.args) != 1:
                raise GoPanic("conversion takes one argument")
            return self.convert(fn.rt, self.eval(e.args[0], env))
        args = [self.eval(a, env) for a in e.args]
        if len(args) == 1 and isinstance(args[0], tuple) and not (args[0] and args[0][0] == "$spread"):
            args = list(args[0])
        if e.ellipsis:
            args[-1] = ("$spread", args[-1])
        return self.call(fn, args)

    def builtin(self, name, e, env):
        a = e.args
        if name == "len" or name == "cap":
            v = self.eval(a[0], env)
            if isinstance(v, GoPtr):
                v = v.v
            if v is None:
                return 0
            if isinstance(v, (str, dict)):
                return len(v)
            return v.c if name == "cap" and isinstance(v, GoSlice) else v.n
        if name == "make":
            tv = self.eval(a[0], env)
            rt = tv.rt
            u = rt.u()
            if u.kind == "slice":
                n = int(self.eval(a[1], env))
                c = int(self.eval(a[2], env)) if len(a) > 2 else n
                if c < n:
                    raise GoPanic("make: len larger than cap")
                z = self.zero(u.elem)
                if isinstance(z, (GoStruct, GoArray)):
                    back = [self.zero(u.elem) for _ in range(c)]
                else:
                    back = [z] * c
                return GoSlice(back, 0, n, c, u.elem)
            if u.kind == "map":
                return {}
            raise Unsupported("make of this type")
        if name == "new":
            tv = self.eval(a[0], env)
            z = self.zero(tv.rt)
            if isinstance(z, (GoStruct, GoArray)):
                return GoPtr(z)
            box = Env()
            box.vars["v"] = z
            return VarPtr(box, "v")
        if name == "append":
            s = self.eval(a[0], env)
            if e.ellipsis:
                src = self.eval(a[1], env)
                vals = [] if src is None else src.a[src.o:src.o + src.n]
                et = s.et if s is not None else src.et
            else:
                vals = [self.eval(x, env) for x in a[1:]]
                et = s.et if s is not None else None
            if s is None:
                if et is None:
                    pkg, imports = self.file_of(env)
                    et = None
                vals = [self.coerce(et, v) for v in vals]
                return GoSlice(list(vals), 0, len(vals), len(vals), et)
            vals = [self.coerce(s.et, v) for v in vals]
            if s.n + len(vals) <= s.c:
                for i, v in enumerate(vals):
                    idx = s.o + s.n + i
                    if idx < len(s.a):
                        s.a[idx] = v
                    else:
                        s.a.append(v)
                return GoSlice(s.a, s.o, s.n + len(vals), s.c, s.et)
            newcap = max(2 * s.c, s.n + len(vals))
            back = s.a[s.o:s.o + s.n] + vals
            z = self.zero(s.et) if s.et is not None else None
            back += [z] * (newcap - len(back))
            return GoSlice(back, 0, s.n + len(vals), newcap, s.et)
        if name == "copy":
            d, s = self.eval(a[0], env), self.eval(a[1], env)
            if d is None or s is None:
                return 0
            n = min(d.n, s.n)
            vals = [self.copy_value(x) for x in s.a[s.o:s.o + n]]
            d.a[d.o:d.o + n] = vals
            return n
        if name == "panic":
            raise GoPanic(str(self.eval(a[0], env)))
        if name == "real":
            return self.eval(a[0], env).real
        if name == "imag":
            return self.eval(a[0], env).imag
        if name == "complex":
            return complex(float(self.eval(a[0], env)), float(self.eval(a[1], env)))
        if name == "delete":
            m = self.eval(a[0], env)
            m.pop(self.eval(a[1], env), None)
            return None
        if name == "recover":
            return None
        if name in ("print", "println"):
            return None
        if name == "min":
            return min(self.eval(x, env) for x in a)
        if name == "max":
            return max(self.eval(x, env) for x in a)
        raise Unsupported(f"builtin {name}")


BUILTINS = {"len", "cap", "make", "new", "append", "copy", "panic", "real", "imag", "complex", "delete", "recover", "print", "println", "min", "max"}


def ptr_load(p):
    if isinstance(p, VarPtr):
        return p.env.vars[p.name]
    if isinstance(p, FieldPtr):
        return p.s.f[p.name]
    if isinstance(p, ElemPtr):
        return p.a[p.i]
    if isinstance(p, GoPtr):
        return p.v
    raise GoPanic("load through a nil pointer")


def ptr_store(p, v):
    if isinstance(p, VarPtr):
        p.env.vars[p.name] = v
    elif isinstance(p, FieldPtr):
        p.s.f[p.name] = v
    elif isinstance(p, ElemPtr):
        p.a[p.i] = v
    else:
        raise GoPanic("store through a nil pointer")


# ------------------------------------------------------------------------------------------------------- numpy bridges

def slice_to_np(s, dtype):
    if s is None:
        return np.zeros(0, dtype)
    return np.array(s.a[s.o:s.o + s.n], dtype=dtype)


def np_to_slice(arr, et, elem):
    vals = [elem(x) for x in arr.tolist()]
    return GoSlice(vals, 0, len(vals), len(vals), et)
