"""gocheck -- a small static checker for the Go side of the boundary (shim/go/**), for an image that has no Go toolchain.

It is NOT a Go compiler.  It parses the subset of Go the shim is written in (and the top-level declarations of the
reference packages under /root/reference), infers the type of every expression, and reports what `go build` would refuse
in the places a cgo shim goes wrong:

  * an identifier `pkg.Name` that the package does not export; a field or method that a type does not have;
  * a call with the wrong number of arguments, or an argument / assignment / return value / composite-literal field /
    append operand whose type is not assignable -- in particular any crossing between []params.Torus (`type Torus uint32`
    is a DEFINED type, params/params.go:27) and []uint32, which is what round 4's shim got wrong three times;
  * a call of a C function whose arguments do not match the prototype in include/tfhe_hip.h;
  * an import that is not used, a package used without an import, a local declared and not used, a missing return count.

What it does not know: the full standard library (a hand-written table of the few names the shim uses), generics, labels,
switch / select / goto, method sets of embedded types beyond sync.Mutex.  Anything outside the subset is an ERROR
("unsupported syntax"), not silently accepted, so the shim has to stay inside what is checked.
"""
import os
import re

# --------------------------------------------------------------------------------------------------------- tokens

KEYWORDS = {"break", "case", "chan", "const", "continue", "default", "defer", "else", "fallthrough", "for", "func", "go", "goto",
            "if", "import", "interface", "map", "package", "range", "return", "select", "struct", "switch", "type", "var"}
OPS = ["<<=", ">>=", "&^=", "...", "&&", "||", "<-", "++", "--", "==", "!=", "<=", ">=", ":=", "+=", "-=", "*=", "/=", "%=", "&=", "|=", "^=",
       "<<", ">>", "&^", "+", "-", "*", "/", "%", "&", "|", "^", "<", ">", "=", "!", "(", ")", "[", "]", "{", "}", ",", ";", ".", ":"]


class GoError(Exception):
    pass


class Tok:
    __slots__ = ("kind", "text", "line")

    def __init__(self, kind, text, line):
        self.kind, self.text, self.line = kind, text, line

    def __repr__(self):
        return f"{self.kind}:{self.text}@{self.line}"


def tokenize(src, fname="<go>"):
    toks, i, line, n = [], 0, 1, len(src)

    def asi():
        if toks:
            t = toks[-1]
            if t.kind in ("ident", "int", "float", "string", "char") or (t.kind == "kw" and t.text in ("break", "continue", "fallthrough", "return")) \
                    or (t.kind == "op" and t.text in ("++", "--", ")", "]", "}")):
                toks.append(Tok("op", ";", line))

    while i < n:
        c = src[i]
        if c == "\n":
            asi()
            line += 1
            i += 1
        elif c in " \t\r":
            i += 1
        elif src.startswith("//", i):
            while i < n and src[i] != "\n":
                i += 1
        elif src.startswith("/*", i):
            j = src.index("*/", i + 2)
            if "\n" in src[i:j]:
                asi()
            line += src.count("\n", i, j)
            i = j + 2
        elif c == '"':
            j = i + 1
            while src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            toks.append(Tok("string", src[i:j + 1], line))
            i = j + 1
        elif c == "`":
            j = src.index("`", i + 1)
            toks.append(Tok("string", src[i:j + 1], line))
            line += src.count("\n", i, j)
            i = j + 1
        elif c == "'":
            j = i + 1
            while src[j] != "'":
                j += 2 if src[j] == "\\" else 1
            toks.append(Tok("char", src[i:j + 1], line))
            i = j + 1
        elif c.isdigit() or (c == "." and i + 1 < n and src[i + 1].isdigit()):
            m = re.match(r"0[xX][0-9a-fA-F_]+|0[oO][0-7_]+|0[bB][01_]+|(\d[\d_]*)?\.\d[\d_]*([eE][+-]?\d+)?|\d[\d_]*[eE][+-]?\d+|\d[\d_]*\.?", src[i:])
            text = m.group(0)
            if text.endswith(".") and src[i + len(text):i + len(text) + 1].isalpha():      # 1.method -- not in the subset
                text = text[:-1]
            toks.append(Tok("float" if re.search(r"[.eE]", text) and not text.lower().startswith(("0x", "0o", "0b")) else "int", text, line))
            i += len(text)
        elif c.isalpha() or c == "_":
            m = re.match(r"[A-Za-z_][A-Za-z_0-9]*", src[i:])
            text = m.group(0)
            toks.append(Tok("kw" if text in KEYWORDS else "ident", text, line))
            i += len(text)
        else:
            for op in OPS:
                if src.startswith(op, i):
                    toks.append(Tok("op", op, line))
                    i += len(op)
                    break
            else:
                raise GoError(f"{fname}:{line}: unexpected character {c!r}")
    asi()
    toks.append(Tok("eof", "", line))
    return toks


# --------------------------------------------------------------------------------------------------------- AST

class Node:
    def __init__(self, kind, line, **kw):
        self.kind, self.line = kind, line
        self.__dict__.update(kw)

    def __repr__(self):
        return f"<{self.kind}@{self.line}>"


BINPREC = {"||": 1, "&&": 2, "==": 3, "!=": 3, "<": 3, "<=": 3, ">": 3, ">=": 3, "+": 4, "-": 4, "|": 4, "^": 4,
           "*": 5, "/": 5, "%": 5, "<<": 5, ">>": 5, "&": 5, "&^": 5}
ASSIGN_OPS = {"=", ":=", "+=", "-=", "*=", "/=", "%=", "&=", "|=", "^=", "<<=", ">>=", "&^="}


class Parser:
    def __init__(self, src, fname):
        self.fname = fname
        self.toks = tokenize(src, fname)
        self.p = 0

    # -- helpers
    @property
    def t(self):
        return self.toks[self.p]

    def err(self, msg, tok=None):
        tok = tok or self.t
        raise GoError(f"{self.fname}:{tok.line}: {msg} (at {tok.text!r})")

    def at(self, text):
        return self.t.kind in ("op", "kw") and self.t.text == text

    def accept(self, text):
        if self.at(text):
            self.p += 1
            return True
        return False

    def expect(self, text):
        if not self.accept(text):
            self.err(f"expected {text!r}")

    def ident(self):
        if self.t.kind != "ident":
            self.err("expected identifier")
        self.p += 1
        return self.toks[self.p - 1].text

    def skip_semis(self):
        while self.accept(";"):
            pass

    # -- file
    def parse_file(self, bodies=True):
        self.skip_semis()
        self.expect("package")
        f = Node("file", 1, package=self.ident(), imports=[], decls=[], fname=self.fname)
        self.skip_semis()
        while self.at("import"):
            self.p += 1
            if self.accept("("):
                self.skip_semis()
                while not self.at(")"):
                    f.imports.append(self.import_spec())
                    self.skip_semis()
                self.expect(")")
            else:
                f.imports.append(self.import_spec())
            self.skip_semis()
        while self.t.kind != "eof":
            if self.at("func"):
                f.decls.append(self.func_decl(bodies))
            elif self.at("type"):
                f.decls.extend(self.gen_decl(self.type_spec))
            elif self.at("var"):
                f.decls.extend(self.gen_decl(lambda: self.value_spec("var")))
            elif self.at("const"):
                f.decls.extend(self.gen_decl(lambda: self.value_spec("const")))
            else:
                self.err("unsupported top-level syntax")
            self.skip_semis()
        return f

    def import_spec(self):
        line = self.t.line
        alias = None
        if self.t.kind == "ident":
            alias = self.ident()
        elif self.accept("."):
            self.err("dot imports are outside the checked subset")
        if self.t.kind != "string":
            self.err("expected import path")
        path = self.t.text[1:-1]
        self.p += 1
        return Node("import", line, alias=alias, path=path)

    def gen_decl(self, spec):
        self.p += 1
        out = []
        if self.accept("("):
            self.skip_semis()
            while not self.at(")"):
                out.append(spec())
                self.skip_semis()
            self.expect(")")
        else:
            out.append(spec())
        return out

    def type_spec(self):
        line = self.t.line
        name = self.ident()
        alias = self.accept("=")
        return Node("typedecl", line, name=name, alias=alias, type=self.type_())

    def value_spec(self, what):
        line = self.t.line
        names = [self.ident()]
        while self.accept(","):
            names.append(self.ident())
        typ = None
        if not self.at("=") and not self.at(";") and not self.at(")"):
            typ = self.type_()
        values = []
        if self.accept("="):
            values = self.expr_list()
        return Node(what, line, names=names, type=typ, values=values)

    def func_decl(self, bodies):
        line = self.t.line
        self.expect("func")
        recv = None
        if self.at("("):
            ps = self.params()
            if len(ps) != 1:
                self.err("method receiver")
            recv = ps[0]
        name = self.ident()
        if self.at("["):                               # type parameters (the reference has one generic helper): skipped, the
            depth = 0                                  # function then fails to resolve and is simply not indexed
            while True:
                if self.at("["):
                    depth += 1
                elif self.at("]"):
                    depth -= 1
                self.p += 1
                if depth == 0:
                    break
        sig = self.signature(line)
        body, lazy = None, None
        if self.at("{"):
            if bodies:
                body = self.block()
            else:
                lazy = (self, self.p)                  # parse_lazy_body() parses it on demand (tools/go_static/gointerp.py)
                self.skip_block()
        return Node("funcdecl", line, name=name, recv=recv, sig=sig, body=body, lazy=lazy)

    def parse_body_at(self, pos):
        saved, self.p = self.p, pos
        try:
            return self.block()
        finally:
            self.p = saved

    def skip_block(self):
        depth = 0
        while True:
            if self.at("{"):
                depth += 1
            elif self.at("}"):
                depth -= 1
                if depth == 0:
                    self.p += 1
                    return
            elif self.t.kind == "eof":
                self.err("unbalanced braces")
            self.p += 1

    # -- types
    def signature(self, line):
        params = self.params()
        results = []
        if self.at("("):
            results = self.params()
        elif not (self.at("{") or self.at(";") or self.at(")") or self.at(",") or self.at("}") or self.at("]") or self.at("=") or self.at(":=") or self.t.kind in ("eof", "string")):
            results = [Node("param", line, name=None, type=self.type_(), variadic=False)]
        return Node("functype", line, params=params, results=results)

    def params(self):
        """( [names] type, ... ) -> list of param nodes (one per name; unnamed: name None)."""
        self.expect("(")
        groups = []                     # each: ([maybe-names-as-types], type-or-None)
        entries = []
        while not self.at(")"):
            line = self.t.line
            variadic = self.accept("...")
            typ = self.type_()
            if not variadic and not self.at(",") and not self.at(")"):
                # `name Type` (the "type" just parsed was really a name)
                if typ.kind != "tname" or typ.pkg is not None:
                    self.err("parameter name")
                variadic2 = self.accept("...")
                entries.append((typ.name, self.type_(), variadic2, line))
            else:
                entries.append((None, typ, variadic, line))
            if not self.accept(","):
                break
        self.expect(")")
        # Go rule: either all entries are named or none; `a, b T` shows up as (None, tname a), (b, T)
        if any(e[0] for e in entries):
            out, pending = [], []
            for name, typ, var, line in entries:
                if name is None:
                    if typ.kind != "tname" or typ.pkg is not None:
                        self.err("mixed named and unnamed parameters")
                    pending.append((typ.name, line))
                else:
                    for pn, pl in pending:
                        out.append(Node("param", pl, name=pn, type=typ, variadic=var))
                    pending = []
                    out.append(Node("param", line, name=name, type=typ, variadic=var))
            if pending:
                self.err("mixed named and unnamed parameters")
            return out
        return [Node("param", line, name=None, type=typ, variadic=var) for _, typ, var, line in entries]

    def type_(self):
        line = self.t.line
        if self.accept("*"):
            return Node("tptr", line, elem=self.type_())
        if self.accept("("):
            t = self.type_()
            self.expect(")")
            return t
        if self.accept("["):
            if self.accept("]"):
                return Node("tslice", line, elem=self.type_())
            if self.t.kind != "int":
                self.err("array length must be an integer literal in the checked subset")
            n = int(self.t.text, 0)
            self.p += 1
            self.expect("]")
            return Node("tarray", line, len=n, elem=self.type_())
        if self.accept("map"):
            self.expect("[")
            k = self.type_()
            self.expect("]")
            return Node("tmap", line, key=k, elem=self.type_())
        if self.accept("func"):
            return self.signature(line)
        if self.accept("struct"):
            return self.struct_body(line)
        if self.accept("interface"):
            self.expect("{")
            self.skip_semis()
            if not self.accept("}"):
                # method sets of reference interfaces are not needed: skip
                depth = 1
                while depth:
                    if self.at("{"):
                        depth += 1
                    elif self.at("}"):
                        depth -= 1
                    self.p += 1
            return Node("tiface", line)
        if self.at("chan"):
            self.err("channels are outside the checked subset")
        if self.t.kind == "ident":
            name = self.ident()
            if self.at(".") and self.toks[self.p + 1].kind == "ident":
                self.p += 1
                return Node("tname", line, pkg=name, name=self.ident())
            return Node("tname", line, pkg=None, name=name)
        self.err("expected a type")

    def struct_body(self, line):
        self.expect("{")
        fields = []
        self.skip_semis()
        while not self.at("}"):
            fl = self.t.line
            if self.at("*"):
                t = self.type_()
                fields.append((None, t))
            else:
                first = self.type_()
                if self.at(";") or self.at("}") or self.t.kind == "string":
                    fields.append((None, first))                      # embedded
                else:
                    if first.kind != "tname" or first.pkg is not None:
                        self.err("field name")
                    names = [first.name]
                    while self.accept(","):
                        names.append(self.ident())
                    t = self.type_()
                    for nm in names:
                        fields.append((nm, t))
            if self.t.kind == "string":
                self.p += 1                                           # tag
            self.skip_semis()
            _ = fl
        self.expect("}")
        return Node("tstruct", line, fields=fields)

    # -- statements
    def block(self):
        line = self.t.line
        self.expect("{")
        stmts = []
        self.skip_semis()
        while not self.at("}"):
            stmts.append(self.stmt())
            self.skip_semis()
        self.expect("}")
        return Node("block", line, stmts=stmts)

    def stmt(self):
        line = self.t.line
        if self.at("{"):
            return self.block()
        if self.at("var"):
            specs = self.gen_decl(lambda: self.value_spec("var"))
            return Node("declstmt", line, specs=specs)
        if self.at("const"):
            specs = self.gen_decl(lambda: self.value_spec("const"))
            return Node("declstmt", line, specs=specs)
        if self.accept("return"):
            vals = [] if (self.at(";") or self.at("}")) else self.expr_list()
            return Node("return", line, values=vals)
        if self.accept("defer"):
            return Node("defer", line, call=self.expr())
        if self.accept("go"):
            return Node("go", line, call=self.expr())
        if self.at("continue") or self.at("break"):
            what = self.t.text
            self.p += 1
            if self.t.kind == "ident":
                self.err("labelled break / continue is outside the checked subset")
            return Node("branch", line, what=what)
        if self.accept("if"):
            return self.if_stmt(line)
        if self.accept("for"):
            return self.for_stmt(line)
        if self.accept("switch"):
            return self.switch_stmt(line)
        for kw in ("select", "goto", "fallthrough", "type"):
            if self.at(kw):
                self.err(f"`{kw}` is outside the checked subset")
        return self.simple_stmt()

    def simple_stmt(self, nolit=False):
        line = self.t.line
        lhs = self.expr_list(nolit)
        if self.t.kind == "op" and self.t.text in ASSIGN_OPS:
            op = self.t.text
            self.p += 1
            if self.accept("range"):
                return Node("rangeassign", line, lhs=lhs, define=op == ":=", x=self.expr(nolit))
            return Node("assign", line, lhs=lhs, op=op, rhs=self.expr_list(nolit))
        if self.at("++") or self.at("--"):
            op = self.t.text
            self.p += 1
            return Node("incdec", line, x=lhs[0], op=op)
        if len(lhs) != 1:
            self.err("expression list is not a statement")
        return Node("exprstmt", line, x=lhs[0])

    def switch_stmt(self, line):
        """switch [init;] [tag] { case a, b: ... default: ... }   (expression switches; used by the interpreter, not by the shim checker)"""
        init = tag = None
        if not self.at("{"):
            first = None if self.at(";") else self.simple_stmt(nolit=True)
            if self.accept(";"):
                init = first
                if not self.at("{"):
                    t = self.simple_stmt(nolit=True)
                    if t.kind != "exprstmt":
                        self.err("switch tag")
                    tag = t.x
            else:
                if first is None or first.kind != "exprstmt":
                    self.err("switch tag")
                tag = first.x
        self.expect("{")
        clauses = []
        self.skip_semis()
        while not self.at("}"):
            cl = self.t.line
            if self.accept("default"):
                exprs = None
            else:
                self.expect("case")
                exprs = self.expr_list()
            self.expect(":")
            body = []
            self.skip_semis()
            while not (self.at("case") or self.at("default") or self.at("}")):
                body.append(self.stmt())
                self.skip_semis()
            clauses.append((exprs, Node("block", cl, stmts=body)))
        self.expect("}")
        return Node("switch", line, init=init, tag=tag, clauses=clauses)

    def if_stmt(self, line):
        init = None
        cond = self.simple_stmt(nolit=True)
        if self.accept(";"):
            init = cond
            cond = self.simple_stmt(nolit=True)
        if cond.kind != "exprstmt":
            self.err("if condition")
        then = self.block()
        els = None
        if self.accept("else"):
            els = self.if_stmt(self.t.line) if self.accept("if") else self.block()
        return Node("if", line, init=init, cond=cond.x, then=then, els=els)

    def for_stmt(self, line):
        if self.at("{"):
            return Node("for", line, init=None, cond=None, post=None, body=self.block())
        if self.accept("range"):
            x = self.expr(nolit=True)
            return Node("forrange", line, lhs=[], define=False, x=x, body=self.block())
        init = cond = post = None
        first = None if self.at(";") else self.simple_stmt(nolit=True)
        if first is not None and first.kind == "rangeassign":
            return Node("forrange", line, lhs=first.lhs, define=first.define, x=first.x, body=self.block())
        if self.at("{"):
            if first.kind != "exprstmt":
                self.err("for condition")
            return Node("for", line, init=None, cond=first.x, post=None, body=self.block())
        init = first
        self.expect(";")
        if not self.at(";"):
            c = self.simple_stmt(nolit=True)
            if c.kind != "exprstmt":
                self.err("for condition")
            cond = c.x
        self.expect(";")
        if not self.at("{"):
            post = self.simple_stmt(nolit=True)
        return Node("for", line, init=init, cond=cond, post=post, body=self.block())

    # -- expressions
    def expr_list(self, nolit=False):
        out = [self.expr(nolit)]
        while self.accept(","):
            out.append(self.expr(nolit))
        return out

    def expr(self, nolit=False, prec=1):
        x = self.unary(nolit)
        while self.t.kind == "op" and self.t.text in BINPREC and BINPREC[self.t.text] >= prec:
            op, line = self.t.text, self.t.line
            self.p += 1
            y = self.expr(nolit, BINPREC[op] + 1)
            x = Node("binary", line, op=op, x=x, y=y)
        return x

    def unary(self, nolit):
        line = self.t.line
        if self.t.kind == "op" and self.t.text in ("+", "-", "!", "^", "*", "&"):
            op = self.t.text
            self.p += 1
            return Node("unary", line, op=op, x=self.unary(nolit))
        if self.at("<-"):
            self.err("channels are outside the checked subset")
        return self.primary(nolit)

    def is_type_start(self):
        return self.at("[") or self.at("map") or self.at("struct") or self.at("interface") or self.at("chan")

    def primary(self, nolit):
        line = self.t.line
        t = self.t
        if t.kind in ("int", "float", "string", "char"):
            self.p += 1
            x = Node("lit", line, lkind=t.kind, text=t.text)
        elif t.kind == "ident":
            self.p += 1
            x = Node("ident", line, name=t.text)
        elif self.at("("):
            self.p += 1
            inner = self.expr()
            self.expect(")")
            x = Node("paren", line, x=inner)
        elif self.at("func"):
            self.p += 1
            sig = self.signature(line)
            if self.at("{"):
                x = Node("funclit", line, sig=sig, body=self.block())
            else:
                x = sig                                   # a function TYPE in expression position (conversion)
        elif self.is_type_start():
            x = self.type_()
        else:
            self.err("unexpected token in expression")
        while True:
            line = self.t.line
            if self.at("."):
                self.p += 1
                if self.at("("):
                    self.p += 1
                    t = self.type_()
                    self.expect(")")
                    x = Node("typeassert", line, x=x, type=t)
                    continue
                x = Node("selector", line, x=x, sel=self.ident())
            elif self.at("("):
                self.p += 1
                args, ell = [], False
                while not self.at(")"):
                    args.append(self.expr())
                    if self.accept("..."):
                        ell = True
                    if not self.accept(","):
                        break
                    self.skip_semis()
                self.expect(")")
                x = Node("call", line, fun=x, args=args, ellipsis=ell)
            elif self.at("["):
                self.p += 1
                parts, colons = [], 0
                cur = None
                while not self.at("]"):
                    if self.accept(":"):
                        parts.append(cur)
                        cur = None
                        colons += 1
                    else:
                        cur = self.expr()
                parts.append(cur)
                self.expect("]")
                x = Node("index", line, x=x, index=parts[0]) if colons == 0 else Node("slice", line, x=x, parts=parts)
            elif self.at("{") and self.lit_type_ok(x) and (not nolit or x.kind in ("tslice", "tarray", "tmap", "tstruct")):
                x = self.composite(x)
            else:
                return x

    @staticmethod
    def lit_type_ok(x):
        if x.kind in ("tslice", "tarray", "tmap", "tstruct"):
            return True
        if x.kind == "ident":
            return True
        return x.kind == "selector" and x.x.kind == "ident"

    def composite(self, typ):
        line = self.t.line
        self.expect("{")
        elts = []
        self.skip_semis()
        while not self.at("}"):
            v = self.lit_value()
            if self.accept(":"):
                elts.append((v, self.lit_value()))
            else:
                elts.append((None, v))
            if not self.accept(","):
                self.skip_semis()
                break
            self.skip_semis()
        self.expect("}")
        return Node("complit", line, type=typ, elts=elts)

    def lit_value(self):
        if self.at("{"):                                  # elided element type: {a, b}
            return self.composite(None)
        return self.expr()


def parse_source(src, fname, bodies=True):
    return Parser(src, fname).parse_file(bodies)


# --------------------------------------------------------------------------------------------------------- types

BASIC = {"int", "int8", "int16", "int32", "int64", "uint", "uint8", "uint16", "uint32", "uint64", "uintptr", "float32", "float64",
         "bool", "string", "byte", "rune", "error", "complex128"}
INTS = {"int", "int8", "int16", "int32", "int64", "uint", "uint8", "uint16", "uint32", "uint64", "uintptr"}
FLOATS = {"float32", "float64"}


def T_basic(n):
    if n == "error":
        return ("iface",)
    return ("basic", {"byte": "uint8", "rune": "int32"}.get(n, n))


T_BOOL, T_INT, T_STRING = T_basic("bool"), T_basic("int"), T_basic("string")
T_NIL, T_UNKNOWN, T_IFACE = ("untyped", "nil"), ("unknown",), ("iface",)
T_UPTR = ("named", "unsafe.Pointer")


def tstr(t):
    k = t[0]
    if k in ("basic", "named"):
        return t[1]
    if k == "untyped":
        return f"untyped {t[1]}"
    if k == "ptr":
        return "*" + tstr(t[1])
    if k == "slice":
        return "[]" + tstr(t[1])
    if k == "array":
        return f"[{t[1]}]" + tstr(t[2])
    if k == "map":
        return f"map[{tstr(t[1])}]{tstr(t[2])}"
    if k == "func":
        return "func(" + ", ".join(tstr(p) for p in t[1]) + ")" + ("" if not t[2] else " (" + ", ".join(tstr(r) for r in t[2]) + ")")
    if k == "tuple":
        return "(" + ", ".join(tstr(x) for x in t[1]) + ")"
    if k == "struct":
        return "struct{" + "; ".join(f"{n} {tstr(ft)}" for n, ft in t[1]) + "}"
    if k == "type":
        return "type " + tstr(t[1])
    if k == "pkg":
        return "package " + t[1]
    return k


class Package:
    """Top-level declarations of one Go package: types (struct fields, aliases, defined types), funcs, methods, vars, consts."""

    def __init__(self, name, path):
        self.name, self.path = name, path
        self.types = {}          # name -> ('alias', T) | ('defined', underlying T)
        self.funcs = {}          # name -> func type
        self.methods = {}        # (typename, method) -> func type (receiver dropped)
        self.values = {}         # var / const name -> type (or T_UNKNOWN)
        self.pending = []


class World:
    """All packages known to the checker: the reference's (by import path), the shim's, a tiny slice of the standard
    library, and "C" built from include/tfhe_hip.h."""

    def __init__(self, lenient_std=False):
        self.by_path = {}
        self.errors = []
        self.lenient_std = lenient_std          # standard-library packages outside the hand-written table: everything they export is
        self._std()                             # of unknown type (accepted anywhere) instead of an error -- for tools/go_golden/main.go

    def is_std(self, path):
        return "." not in path.split("/", 1)[0] and path != "C"

    # ---- resolving syntactic types
    def resolve_type(self, node, pkg, imports):
        k = node.kind
        if k == "tname":
            if node.pkg is None:
                if node.name in pkg.types or node.name in getattr(pkg, "_declared_types", ()):
                    return ("named", f"{pkg.path}.{node.name}")
                if node.name in BASIC:
                    return T_basic(node.name)
                if node.name == "any":
                    return T_IFACE
                raise GoError(f"{pkg.path}: unknown type {node.name} (line {node.line})")
            path = imports.get(node.pkg)
            if path is None:
                raise GoError(f"{pkg.path}: package {node.pkg} is not imported (line {node.line})")
            target = self.by_path.get(path)
            if target is None and self.lenient_std and self.is_std(path):
                return T_UNKNOWN
            if target is None:
                raise GoError(f"{pkg.path}: import {path!r} is not known to the checker (line {node.line})")
            if path == "C":
                if node.name not in target.types:
                    raise GoError(f"C.{node.name} is not declared by include/tfhe_hip.h (line {node.line})")
            elif node.name not in target.types and node.name not in getattr(target, "_declared_types", ()):
                raise GoError(f"{path} has no type {node.name} (line {node.line})")
            elif not node.name[0].isupper() and path != "C":
                raise GoError(f"{path}.{node.name} is not exported (line {node.line})")
            return ("named", f"{path}.{node.name}")
        if k == "tptr":
            return ("ptr", self.resolve_type(node.elem, pkg, imports))
        if k == "tslice":
            return ("slice", self.resolve_type(node.elem, pkg, imports))
        if k == "tarray":
            return ("array", node.len, self.resolve_type(node.elem, pkg, imports))
        if k == "tmap":
            return ("map", self.resolve_type(node.key, pkg, imports), self.resolve_type(node.elem, pkg, imports))
        if k == "functype":
            ps = tuple(self.resolve_type(p.type, pkg, imports) for p in node.params)
            rs = tuple(self.resolve_type(r.type, pkg, imports) for r in node.results)
            variadic = bool(node.params and node.params[-1].variadic)
            return ("func", ps, rs, variadic)
        if k == "tstruct":
            return ("struct", tuple((n, self.resolve_type(t, pkg, imports)) for n, t in node.fields))
        if k == "tiface":
            return T_IFACE
        raise GoError(f"{pkg.path}: unsupported type syntax {k} (line {node.line})")

    def named_info(self, t):
        """('alias'|'defined', T) of a named type."""
        path, name = t[1].rsplit(".", 1)
        p = self.by_path.get(path)
        if p is None or name not in p.types:
            return None
        return p.types[name]

    def dealias(self, t):
        while t[0] == "named":
            info = self.named_info(t)
            if info is None or info[0] != "alias":
                return t
            t = info[1]
        k = t[0]
        if k in ("ptr", "slice"):
            return (k, self.dealias(t[1]))
        if k == "array":
            return ("array", t[1], self.dealias(t[2]))
        if k == "map":
            return ("map", self.dealias(t[1]), self.dealias(t[2]))
        if k == "func":
            return ("func", tuple(self.dealias(x) for x in t[1]), tuple(self.dealias(x) for x in t[2]), t[3])
        return t

    def underlying(self, t):
        t = self.dealias(t)
        seen = 0
        while t[0] == "named" and seen < 20:
            info = self.named_info(t)
            if info is None:
                return t
            t = self.dealias(info[1])
            seen += 1
        return t

    def identical(self, a, b):
        return self.dealias(a) == self.dealias(b)

    def assignable(self, v, T):
        """Go assignability of a value of type v to a variable of type T (spec: Assignability), untyped constants included."""
        if v == T_UNKNOWN or T == T_UNKNOWN:
            return True
        v, T = self.dealias(v), self.dealias(T)
        if v == T:
            return True
        uT = self.underlying(T)
        if uT == T_IFACE:
            return True
        if v[0] == "untyped":
            kind = v[1]
            if kind == "nil":
                return uT[0] in ("ptr", "slice", "map", "func") or uT == T_IFACE or T == T_UPTR
            if uT[0] != "basic":
                return False
            b = uT[1]
            if kind in ("int", "rune"):
                return b in INTS or b in FLOATS or b == "complex128"
            if kind == "float":
                return b in FLOATS or b == "complex128"
            if kind == "bool":
                return b == "bool"
            if kind == "string":
                return b == "string"
            return False
        # identical underlying types and at least one of them is not a named type
        if (v[0] != "named" or T[0] != "named") and self.underlying(v) == uT and v[0] != "basic" and T[0] != "basic":
            return True
        return False

    # ---- loading packages
    def load_package(self, path, files, name_hint=None, bodies=False):
        """files: list of (fname, source).  Returns (Package, [file ASTs])."""
        asts = [parse_source(src, fname, bodies) for fname, src in files]
        names = {a.package for a in asts if not a.package.endswith("_test")}
        pkg = Package((names.pop() if names else name_hint) or name_hint, path)
        self.by_path[path] = pkg
        pkg._declared_types = {d.name for a in asts for d in a.decls if d.kind == "typedecl" and not a.package.endswith("_test")}
        pkg._asts = asts
        return pkg, asts

    def resolve_package(self, pkg, strict=True):
        """Second pass (after every package of the world is loaded): resolve declared types / signatures.  strict=False (the
        reference's packages): a declaration outside the subset is left out of the index instead of being reported."""
        for a in pkg._asts:
            if a.package.endswith("_test"):
                continue
            imports = self.import_map(a)
            for d in a.decls:
                try:
                    if d.kind == "typedecl":
                        pkg.types[d.name] = ("alias" if d.alias else "defined", self.resolve_type(d.type, pkg, imports))
                    elif d.kind == "funcdecl":
                        ft = self.resolve_type(d.sig, pkg, imports)
                        if d.recv is None:
                            pkg.funcs[d.name] = ft
                        else:
                            rt = d.recv.type
                            base = rt.elem if rt.kind == "tptr" else rt
                            pkg.methods[(base.name, d.name)] = ft
                    elif d.kind in ("var", "const"):
                        t = self.resolve_type(d.type, pkg, imports) if d.type is not None else T_UNKNOWN
                        for n in d.names:
                            pkg.values[n] = t
                except GoError as e:
                    if strict:
                        self.errors.append(str(e))

    @staticmethod
    def import_map(ast):
        m = {}
        for imp in ast.imports:
            m[imp.alias or imp.path.rsplit("/", 1)[-1]] = imp.path
        return m

    # ---- the slice of the standard library the shim uses
    def _std(self):
        def pk(path, name=None):
            p = Package(name or path.rsplit("/", 1)[-1], path)
            p._declared_types, p._asts = set(), []
            self.by_path[path] = p
            return p
        f = lambda ps, rs=(), var=False: ("func", tuple(ps), tuple(rs), var)      # noqa: E731
        sync = pk("sync")
        sync.types["WaitGroup"] = ("defined", ("struct", ()))
        sync.types["Mutex"] = ("defined", ("struct", ()))
        sync.methods[("WaitGroup", "Add")] = f([T_INT])
        sync.methods[("WaitGroup", "Done")] = f([])
        sync.methods[("WaitGroup", "Wait")] = f([])
        sync.methods[("Mutex", "Lock")] = f([])
        sync.methods[("Mutex", "Unlock")] = f([])
        rt = pk("runtime")
        rt.funcs["LockOSThread"] = f([])
        rt.funcs["UnlockOSThread"] = f([])
        at = pk("sync/atomic", "atomic")
        at.funcs["AddUint32"] = f([("ptr", T_basic("uint32")), T_basic("uint32")], [T_basic("uint32")])
        un = pk("unsafe")
        un.types["Pointer"] = ("defined", ("basic", "uintptr"))
        tst = pk("testing")
        tst.types["T"] = ("defined", ("struct", ()))
        for m in ("Errorf", "Fatalf", "Logf"):
            tst.methods[("T", m)] = f([T_STRING, T_IFACE], [], True)
        tst.methods[("T", "Run")] = f([T_STRING, ("func", (("ptr", ("named", "testing.T")),), (), False)], [T_BOOL])

    # ---- "C": the prototypes of include/tfhe_hip.h as cgo presents them
    def load_c_header(self, text):
        c = Package("C", "C")
        c._declared_types, c._asts = set(), []
        self.by_path["C"] = c
        for n in ("int", "uint", "char", "double", "float", "size_t", "int32_t", "uint32_t", "uint64_t", "uint8_t", "int64_t", "long"):
            c.types[n] = ("defined", T_basic({"int": "int32", "uint": "uint32", "char": "int8", "double": "float64", "float": "float32", "size_t": "uint64",
                                              "int32_t": "int32", "uint32_t": "uint32", "uint64_t": "uint64", "uint8_t": "uint8", "int64_t": "int64",
                                              "long": "int64"}[n]))
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
        text = re.sub(r"//[^\n]*", " ", text)
        # structs: typedef struct { fields } name;   and opaque   typedef struct tag name;
        for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
            fields = []
            for fm in re.finditer(r"(\w+)\s+(\w+)\s*;", m.group(1)):
                fields.append((fm.group(2), ("named", "C." + fm.group(1))))
            c.types[m.group(2)] = ("defined", ("struct", tuple(fields)))
        for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", text):
            c.types[m.group(2)] = ("defined", ("struct", ()))
        for m in re.finditer(r"enum\s*\{(.*?)\}", text, flags=re.S):
            for em in re.finditer(r"(\w+)\s*(=\s*[-\w]+)?\s*(,|$)", m.group(1).strip()):
                c.values[em.group(1)] = ("untyped", "int")
        for m in re.finditer(r"#define\s+(\w+)\s+(-?\d+)", text):
            c.values[m.group(1)] = ("untyped", "int")

        def ctype(s):
            s = s.replace("const", " ").strip()
            stars = s.count("*")
            base = s.replace("*", " ").split()
            if not base:
                raise GoError(f"C prototype: cannot read type {s!r}")
            b = base[0] if base[0] != "unsigned" else "uint"
            if b == "void":
                if stars == 0:
                    return None
                t = T_UPTR
                stars -= 1
            else:
                if b not in c.types:
                    raise GoError(f"C prototype: unknown type {b!r}")
                t = ("named", "C." + b)
            for _ in range(stars):
                t = ("ptr", t)
            return t

        for m in re.finditer(r"\b(int|const\s+char\s*\*|void)\s*\b(tfhe_\w+)\s*\(([^)]*)\)\s*;", text):
            ret, name, args = m.group(1), m.group(2), m.group(3).strip()
            ps = []
            if args and args != "void":
                for a in args.split(","):
                    a = a.strip()
                    am = re.match(r"(.*?)(\w+)$", a, flags=re.S)          # drop the parameter name
                    ps.append(ctype(am.group(1) if am and am.group(1).strip() else a))
            r = ctype(ret)
            c.funcs[name] = ("func", tuple(ps), (r,) if r else (), False)
        c.funcs["GoString"] = ("func", (("ptr", ("named", "C.char")),), (T_STRING,), False)
        return c


# --------------------------------------------------------------------------------------------------------- checker

class Scope:
    def __init__(self, parent=None):
        self.parent, self.vars, self.used, self.decl_line = parent, {}, set(), {}

    def lookup(self, name):
        s = self
        while s is not None:
            if name in s.vars:
                s.used.add(name)
                return s.vars[name]
            s = s.parent
        return None

    def declare(self, name, typ, line):
        if name == "_":
            return
        self.vars[name] = typ
        self.decl_line[name] = line


class Checker:
    def __init__(self, world, pkg, ast):
        self.w, self.pkg, self.ast = world, pkg, ast
        self.imports = World.import_map(ast)
        self.used_imports = set()
        self.errors = []
        self.results = None
        self.in_test = ast.package.endswith("_test")

    def err(self, node, msg):
        self.errors.append(f"{self.ast.fname}:{node.line}: {msg}")

    # ---- entry
    def check(self):
        for d in self.ast.decls:
            if d.kind == "funcdecl" and d.body is not None:
                self.check_func(d)
            elif d.kind in ("var", "const"):
                top = Scope()
                for v in d.values:
                    self.expr(v, top)
                if d.kind == "var" and d.type is None and d.values:
                    # type of the initialiser becomes the variable's type (package-level var X = expr)
                    ts = [self.expr(v, top) for v in d.values]
                    for n, t in zip(d.names, ts):
                        self.pkg.values[n] = self.default_type(t)
            elif d.kind == "typedecl":
                self.touch_type(d.type)
        for alias, path in self.imports.items():
            if alias not in self.used_imports and path != "C":
                self.errors.append(f"{self.ast.fname}: imported and not used: {path!r}")
        return self.errors

    def touch_type(self, node):
        """Mark the packages a syntactic type mentions as used, and resolve it."""
        try:
            t = self.w.resolve_type(node, self.pkg, self.imports)
        except GoError as e:
            self.errors.append(f"{self.ast.fname}: {e}")
            return T_UNKNOWN
        self._mark(node)
        return t

    def _mark(self, node):
        if node is None or not isinstance(node, Node):
            return
        if node.kind == "tname" and node.pkg:
            self.used_imports.add(node.pkg)
        for v in node.__dict__.values():
            if isinstance(v, Node):
                self._mark(v)
            elif isinstance(v, (list, tuple)):
                for x in v:
                    if isinstance(x, Node):
                        self._mark(x)
                    elif isinstance(x, tuple):
                        for y in x:
                            if isinstance(y, Node):
                                self._mark(y)

    def check_func(self, d):
        scope = Scope()
        if d.recv is not None:
            scope.declare(d.recv.name or "_", self.touch_type(d.recv.type), d.line)
            scope.used.add(d.recv.name)
        for p in d.sig.params:
            t = self.touch_type(p.type)
            if p.variadic:
                t = ("slice", t)
            scope.declare(p.name or "_", t, d.line)
            scope.used.add(p.name)                   # parameters may go unused
        saved = self.results
        self.results = [self.touch_type(r.type) for r in d.sig.results]
        self.block(d.body, scope)
        if self.results and not self.terminates(d.body):
            self.err(d, f"missing return at the end of {d.name}")
        self.results = saved

    def terminates(self, block):
        if not block.stmts:
            return False
        last = block.stmts[-1]
        if last.kind == "return":
            return True
        if last.kind == "exprstmt" and last.x.kind == "call" and last.x.fun.kind == "ident" and last.x.fun.name == "panic":
            return True
        if last.kind == "if" and last.els is not None:
            els_ok = self.terminates(last.els) if last.els.kind == "block" else self.terminates(Node("block", 0, stmts=[last.els]))
            return self.terminates(last.then) and els_ok
        if last.kind == "for" and last.cond is None:
            return True
        if last.kind == "block":
            return self.terminates(last)
        return False

    # ---- statements
    def block(self, b, scope, own_scope=True):
        s = Scope(scope) if own_scope else scope
        for st in b.stmts:
            self.stmt(st, s)
        if own_scope:
            self.report_unused(s)

    def report_unused(self, s):
        for name, line in s.decl_line.items():
            if name not in s.used:
                self.errors.append(f"{self.ast.fname}:{line}: {name} declared and not used")

    def stmt(self, st, scope):
        k = st.kind
        if k == "block":
            self.block(st, scope)
        elif k == "exprstmt":
            self.expr(st.x, scope)
        elif k == "declstmt":
            for sp in st.specs:
                vals = [self.expr(v, scope) for v in sp.values]
                if sp.type is not None:
                    t = self.touch_type(sp.type)
                    for v, n in zip(vals, sp.values):
                        self.need_assignable(n, v, t, "variable initialiser")
                    for nme in sp.names:
                        scope.declare(nme, t, st.line)
                else:
                    vals = self.spread(vals, len(sp.names), st)
                    for nme, v in zip(sp.names, vals):
                        # an untyped constant stays untyped (`const m = 32` is usable as any integer type)
                        scope.declare(nme, v if sp.kind == "const" else self.default_type(v), st.line)
        elif k == "assign":
            self.assign(st, scope)
        elif k == "incdec":
            self.expr(st.x, scope)
        elif k == "return":
            vals = [self.expr(v, scope) for v in st.values]
            if len(vals) == 1 and vals[0][0] == "tuple":
                vals = list(vals[0][1])
            if len(vals) != len(self.results or []):
                self.err(st, f"return has {len(vals)} value(s), the function returns {len(self.results or [])}")
            else:
                for v, r, n in zip(vals, self.results, st.values if len(st.values) == len(vals) else [st] * len(vals)):
                    self.need_assignable(n, v, r, "return value")
        elif k in ("defer", "go"):
            if st.call.kind != "call":
                self.err(st, f"{k} needs a function call")
            self.expr(st.call, scope)
        elif k == "branch":
            pass
        elif k == "if":
            s = Scope(scope)
            if st.init is not None:
                self.stmt(st.init, s)
            c = self.expr(st.cond, s)
            if not self.is_bool(c):
                self.err(st, f"non-boolean condition in if statement ({tstr(c)})")
            self.block(st.then, s)
            if st.els is not None:
                self.stmt(st.els, s)
            self.report_unused(s)
        elif k == "for":
            s = Scope(scope)
            if st.init is not None:
                self.stmt(st.init, s)
            if st.cond is not None:
                c = self.expr(st.cond, s)
                if not self.is_bool(c):
                    self.err(st, "non-boolean condition in for statement")
            if st.post is not None:
                self.stmt(st.post, s)
            self.block(st.body, s)
            self.report_unused(s)
        elif k == "forrange":
            s = Scope(scope)
            xt = self.expr(st.x, s)
            ut = self.w.underlying(xt)
            if ut[0] == "slice":
                kt, vt = T_INT, ut[1]
            elif ut[0] == "array":
                kt, vt = T_INT, ut[2]
            elif ut[0] == "map":
                kt, vt = ut[1], ut[2]
            elif ut == T_STRING:
                kt, vt = T_INT, T_basic("rune")
            elif ut == T_UNKNOWN:
                kt = vt = T_UNKNOWN
            else:
                self.err(st, f"cannot range over {tstr(xt)}")
                kt = vt = T_UNKNOWN
            for lhs, t in zip(st.lhs, (kt, vt)):
                if st.define:
                    if lhs.kind != "ident":
                        self.err(st, "range variable")
                    else:
                        s.declare(lhs.name, t, st.line)
                else:
                    self.need_assignable(lhs, t, self.expr(lhs, s), "range assignment")
            self.block(st.body, s)
            self.report_unused(s)
        elif k == "rangeassign":
            self.err(st, "range outside a for statement")
        else:
            self.err(st, f"unsupported statement {k}")

    def spread(self, vals, n, node):
        if len(vals) == 1 and vals[0][0] == "tuple":
            vals = list(vals[0][1])
        if len(vals) == 1 and vals[0] == T_UNKNOWN and n > 1:      # a call into an un-modelled package: as many results as needed
            vals = [T_UNKNOWN] * n
        if len(vals) != n:
            self.err(node, f"assignment mismatch: {n} variable(s) but {len(vals)} value(s)")
            vals = (vals + [T_UNKNOWN] * n)[:n]
        return vals

    def assign(self, st, scope):
        # comma-ok map lookup:  v, ok := m[k]
        if len(st.lhs) == 2 and len(st.rhs) == 1 and st.rhs[0].kind == "index":
            mt = self.w.underlying(self.expr(st.rhs[0].x, scope))
            if mt[0] == "map":
                self.expr(st.rhs[0].index, scope)
                vals = [mt[2], T_BOOL]
            else:
                vals = self.spread([self.expr(st.rhs[0], scope)], 2, st)
        else:
            vals = self.spread([self.expr(r, scope) for r in st.rhs], len(st.lhs), st)
        if st.op == ":=":
            fresh = 0
            for lhs, v in zip(st.lhs, vals):
                if lhs.kind != "ident":
                    self.err(st, "non-name on the left of :=")
                    continue
                if lhs.name != "_" and lhs.name not in scope.vars:
                    fresh += 1
                    if v == T_NIL:
                        self.err(st, "use of untyped nil in a short variable declaration")
                    scope.declare(lhs.name, self.default_type(v), st.line)
                elif lhs.name != "_":
                    self.need_assignable(lhs, v, scope.vars[lhs.name], "assignment")
            if fresh == 0 and not all(l.kind == "ident" and l.name == "_" for l in st.lhs):
                self.err(st, "no new variables on the left of :=")
            return
        for lhs, v, r in zip(st.lhs, vals, st.rhs if len(st.rhs) == len(vals) else [st] * len(vals)):
            if lhs.kind == "ident" and lhs.name == "_":
                continue
            if lhs.kind == "ident":                       # a plain assignment is not a "use"
                s, found = scope, None
                while s is not None and found is None:
                    found = s.vars.get(lhs.name)
                    s = s.parent
                lt = found if found is not None else self.expr(lhs, scope)
            else:
                lt = self.expr(lhs, scope)
            if st.op == "=":
                self.need_assignable(r, v, lt, "assignment")
            else:
                self.binary_result(st, st.op[:-1], lt, v)

    # ---- expressions
    def default_type(self, t):
        if t[0] == "untyped":
            return {"int": T_INT, "float": T_basic("float64"), "bool": T_BOOL, "string": T_STRING, "rune": T_basic("rune")}.get(t[1], t)
        return t

    def is_bool(self, t):
        return t == T_UNKNOWN or t == ("untyped", "bool") or self.w.underlying(t) == T_BOOL

    def need_assignable(self, node, v, T, what):
        if v[0] == "type":
            self.err(node, f"{tstr(v[1])} is a type, not a value ({what})")
        elif not self.w.assignable(v, T):
            self.err(node, f"cannot use {tstr(self.w.dealias(v))} as {tstr(self.w.dealias(T))} in {what}")

    def expr(self, e, scope):
        t = self._expr(e, scope)
        return t

    def value(self, e, scope):
        t = self.expr(e, scope)
        if t[0] == "type":
            self.err(e, f"{tstr(t[1])} is a type, not a value")
            return T_UNKNOWN
        if t[0] == "pkg":
            self.err(e, f"use of package {t[1]} without a selector")
            return T_UNKNOWN
        return t

    def _expr(self, e, scope):
        k = e.kind
        if k == "lit":
            return ("untyped", {"int": "int", "float": "float", "string": "string", "char": "rune"}[e.lkind])
        if k == "ident":
            return self.ident(e, scope)
        if k == "paren":
            if e.x.kind.startswith("t") and e.x.kind in ("tptr", "tslice", "tarray", "tmap", "tstruct", "tiface", "tname", "functype"):
                return ("type", self.touch_type(e.x))
            return self.expr(e.x, scope)
        if k in ("tslice", "tarray", "tmap", "tstruct", "tiface", "functype", "tptr", "tname"):
            return ("type", self.touch_type(e))
        if k == "selector":
            return self.selector(e, scope)
        if k == "unary":
            return self.unary(e, scope)
        if k == "binary":
            return self.binary_result(e, e.op, self.value(e.x, scope), self.value(e.y, scope))
        if k == "index":
            xt = self.value(e.x, scope)
            it = self.value(e.index, scope)
            ut = self.w.underlying(xt)
            if ut[0] == "ptr" and self.w.underlying(ut[1])[0] == "array":
                ut = self.w.underlying(ut[1])
            if ut[0] == "map":
                self.need_assignable(e.index, it, ut[1], "map index")
                return ut[2]
            if not self.is_integer(it):
                self.err(e, f"index of type {tstr(it)} is not an integer")
            if ut[0] == "slice":
                return ut[1]
            if ut[0] == "array":
                return ut[2]
            if ut == T_STRING:
                return T_basic("uint8")
            if ut != T_UNKNOWN:
                self.err(e, f"cannot index {tstr(xt)}")
            return T_UNKNOWN
        if k == "slice":
            xt = self.value(e.x, scope)
            for p in e.parts:
                if p is not None and not self.is_integer(self.value(p, scope)):
                    self.err(e, "slice index is not an integer")
            ut = self.w.underlying(xt)
            if ut[0] == "slice":
                return xt if self.w.dealias(xt)[0] == "slice" else xt
            if ut[0] == "array":
                return ("slice", ut[2])
            if ut == T_STRING:
                return xt
            if ut != T_UNKNOWN:
                self.err(e, f"cannot slice {tstr(xt)}")
            return T_UNKNOWN
        if k == "call":
            return self.call(e, scope)
        if k == "complit":
            return self.complit(e, scope, None)
        if k == "funclit":
            ft = self.touch_type(e.sig)
            s = Scope(scope)
            for p in e.sig.params:
                t = self.touch_type(p.type)
                s.declare(p.name or "_", ("slice", t) if p.variadic else t, e.line)
                s.used.add(p.name)
            saved = self.results
            self.results = [self.touch_type(r.type) for r in e.sig.results]
            self.block(e.body, s)
            if self.results and not self.terminates(e.body):
                self.err(e, "missing return at the end of a function literal")
            self.results = saved
            return ft
        self.err(e, f"unsupported expression {k}")
        return T_UNKNOWN

    def is_integer(self, t):
        if t == T_UNKNOWN:
            return True
        if t[0] == "untyped":
            return t[1] in ("int", "rune")
        u = self.w.underlying(t)
        return u[0] == "basic" and u[1] in INTS

    def is_numeric(self, t):
        if t == T_UNKNOWN:
            return True
        if t[0] == "untyped":
            return t[1] in ("int", "rune", "float")
        u = self.w.underlying(t)
        return u[0] == "basic" and (u[1] in INTS or u[1] in FLOATS)

    def ident(self, e, scope):
        n = e.name
        v = scope.lookup(n)
        if v is not None:
            return v
        if n in ("true", "false"):
            return ("untyped", "bool")
        if n == "nil":
            return T_NIL
        if n == "iota":
            return ("untyped", "int")
        if n in self.pkg.values:
            return self.pkg.values[n]
        if n in self.pkg.funcs:
            return self.pkg.funcs[n]
        if n in self.pkg.types:
            return ("type", ("named", f"{self.pkg.path}.{n}"))
        if n in BASIC:
            return ("type", T_basic(n))
        if n in self.imports:
            self.used_imports.add(n)
            return ("pkg", self.imports[n])
        if n in ("len", "cap", "append", "make", "new", "copy", "panic", "recover", "delete", "print", "println"):
            return ("builtin", n)
        self.err(e, f"undefined: {n}")
        return T_UNKNOWN

    def selector(self, e, scope):
        xt = self.expr(e.x, scope)
        if xt[0] == "pkg":
            path = xt[1]
            p = self.w.by_path.get(path)
            if p is None and self.w.lenient_std and self.w.is_std(path):
                return T_UNKNOWN
            if p is None:
                self.err(e, f"package {path!r} is not known to the checker")
                return T_UNKNOWN
            name = e.sel
            if path != "C" and not name[0].isupper():
                self.err(e, f"{p.name}.{name} is not exported")
            if name in p.funcs:
                return p.funcs[name]
            if name in p.types:
                return ("type", ("named", f"{path}.{name}"))
            if name in p.values:
                return p.values[name]
            self.err(e, f"undefined: {p.name}.{name} ({path} declares no such function, type, variable or constant)")
            return T_UNKNOWN
        if xt[0] == "type":
            self.err(e, "method expressions are outside the checked subset")
            return T_UNKNOWN
        return self.member(e, xt, e.sel)

    def member(self, e, xt, name):
        """Field or method `name` of a value of type xt (one automatic dereference; embedded fields one level deep)."""
        if xt == T_UNKNOWN:
            return T_UNKNOWN
        t = self.w.dealias(xt)
        if t[0] == "ptr":
            t = self.w.dealias(t[1])
        if t == T_UNKNOWN:
            return T_UNKNOWN
        own_pkg = None
        if t[0] == "named":
            path, tn = t[1].rsplit(".", 1)
            p = self.w.by_path.get(path)
            own_pkg = path
            if p is not None and (tn, name) in p.methods:
                if path != self.pkg.path and not name[0].isupper():
                    self.err(e, f"{t[1]}.{name} is not exported")
                return p.methods[(tn, name)]
        st = self.w.underlying(t)
        if st[0] == "struct":
            for fn, ft in st[1]:
                if fn == name:
                    if own_pkg not in (None, self.pkg.path, "C") and not name[0].isupper():
                        self.err(e, f"field {name} of {tstr(t)} is not exported")
                    return ft
            for fn, ft in st[1]:                              # embedded (promoted) members
                if fn is None:
                    ftd = self.w.dealias(ft)
                    base = ftd[1] if ftd[0] == "ptr" else ftd
                    if base[0] == "named" and base[1].rsplit(".", 1)[1] == name:
                        return ft
                    sub = self._member_quiet(ft, name)
                    if sub is not None:
                        return sub
        self.err(e, f"{tstr(self.w.dealias(xt))} has no field or method {name}")
        return T_UNKNOWN

    def _member_quiet(self, xt, name):
        t = self.w.dealias(xt)
        if t[0] == "ptr":
            t = self.w.dealias(t[1])
        if t[0] == "named":
            path, tn = t[1].rsplit(".", 1)
            p = self.w.by_path.get(path)
            if p is not None and (tn, name) in p.methods:
                return p.methods[(tn, name)]
        st = self.w.underlying(t)
        if st[0] == "struct":
            for fn, ft in st[1]:
                if fn == name:
                    return ft
        return None

    def unary(self, e, scope):
        if e.op == "*":
            xt = self.expr(e.x, scope)
            if xt[0] == "type":
                return ("type", ("ptr", xt[1]))
            ut = self.w.underlying(xt)
            if ut[0] == "ptr":
                return ut[1]
            if xt != T_UNKNOWN:
                self.err(e, f"cannot dereference {tstr(xt)}")
            return T_UNKNOWN
        if e.op == "&":
            if e.x.kind == "complit":
                return ("ptr", self.complit(e.x, scope, None))
            xt = self.value(e.x, scope)
            if e.x.kind not in ("ident", "selector", "index", "paren"):
                self.err(e, "cannot take the address of this expression")
            return ("ptr", xt)
        xt = self.value(e.x, scope)
        if e.op == "!":
            if not self.is_bool(xt):
                self.err(e, f"operator ! on {tstr(xt)}")
            return xt if xt[0] == "untyped" else xt
        if e.op in ("-", "+"):
            if not self.is_numeric(xt):
                self.err(e, f"operator {e.op} on {tstr(xt)}")
            return xt
        if e.op == "^":
            if not self.is_integer(xt):
                self.err(e, f"operator ^ on {tstr(xt)}")
            return xt
        self.err(e, f"unsupported unary operator {e.op}")
        return T_UNKNOWN

    def binary_result(self, e, op, a, b):
        if a == T_UNKNOWN or b == T_UNKNOWN:
            return T_BOOL if op in ("==", "!=", "<", "<=", ">", ">=", "&&", "||") else T_UNKNOWN
        if op in ("<<", ">>"):
            if not self.is_integer(b):
                self.err(e, f"shift count of type {tstr(b)}")
            if not self.is_integer(a) and not (a[0] == "untyped" and a[1] == "float"):
                self.err(e, f"shift of {tstr(a)}")
            return a
        if op in ("&&", "||"):
            if not (self.is_bool(a) and self.is_bool(b)):
                self.err(e, f"operator {op} on {tstr(a)} and {tstr(b)}")
            return T_BOOL if a[0] != "untyped" or b[0] != "untyped" else ("untyped", "bool")
        # operands must be identical types, or one an untyped constant convertible to the other
        ua, ub = a[0] == "untyped", b[0] == "untyped"
        if ua and ub:
            res = a if a[1] == "float" or b[1] != "float" else b
            if a[1] == "nil" or b[1] == "nil":
                res = a
        elif ua:
            if not self.w.assignable(a, b):
                self.err(e, f"mismatched types {tstr(a)} and {tstr(self.w.dealias(b))} in operator {op}")
            res = b
        elif ub:
            if not self.w.assignable(b, a):
                self.err(e, f"mismatched types {tstr(self.w.dealias(a))} and {tstr(b)} in operator {op}")
            res = a
        else:
            if not self.w.identical(a, b):
                self.err(e, f"mismatched types {tstr(self.w.dealias(a))} and {tstr(self.w.dealias(b))} in operator {op}")
            res = a
        if op in ("==", "!=", "<", "<=", ">", ">="):
            return ("untyped", "bool") if ua and ub else T_BOOL
        if op == "+" and (self.w.underlying(res) == T_STRING or res == ("untyped", "string")):
            return res
        if op in ("%", "&", "|", "^", "&^"):
            if not self.is_integer(res):
                self.err(e, f"operator {op} on {tstr(res)}")
        elif not self.is_numeric(res):
            self.err(e, f"operator {op} on {tstr(self.w.dealias(res))}")
        return res

    def complit(self, e, scope, elided):
        if e.type is None:
            if elided is None:
                self.err(e, "composite literal without a type")
                return T_UNKNOWN
            t = elided
        else:
            tt = self.expr(e.type, scope)
            if tt[0] != "type":
                self.err(e, f"{tstr(tt)} is not a type (composite literal)")
                return T_UNKNOWN
            t = tt[1]
        ut = self.w.underlying(t)
        if ut[0] == "ptr" and e.type is None:                 # elided &T{} inside []*T{ {...} }
            inner = self.complit(Node("complit", e.line, type=None, elts=e.elts), scope, ut[1])
            return ("ptr", inner)
        if ut[0] == "struct":
            fields = dict((n, ft) for n, ft in ut[1] if n is not None)
            embedded = [ft for n, ft in ut[1] if n is None]
            keyed = [kk for kk, _ in e.elts if kk is not None]
            if keyed and len(keyed) != len(e.elts):
                self.err(e, "mixture of field:value and value elements in a struct literal")
            if keyed:
                seen = set()
                for kk, v in e.elts:
                    if kk is None:
                        continue
                    if kk.kind != "ident":
                        self.err(e, "struct literal key is not a field name")
                        continue
                    if kk.name in seen:
                        self.err(e, f"duplicate field {kk.name} in a struct literal")
                    seen.add(kk.name)
                    ft = fields.get(kk.name)
                    if ft is None:
                        emb = [x for x in embedded if self.w.dealias(x)[-1].rsplit(".", 1)[-1] == kk.name] if embedded else []
                        if emb:
                            ft = emb[0]
                        else:
                            self.err(e, f"unknown field {kk.name} in a struct literal of type {tstr(self.w.dealias(t))}")
                            self.elt(v, scope, T_UNKNOWN)
                            continue
                    if t[0] == "named" and not t[1].startswith(self.pkg.path + ".") and not t[1].startswith("C.") and not kk.name[0].isupper():
                        self.err(e, f"field {kk.name} of {tstr(t)} is not exported")
                    self.need_assignable(v, self.elt(v, scope, ft), ft, f"field {kk.name}")
            else:
                if e.elts and len(e.elts) != len(ut[1]):
                    self.err(e, f"too few or too many values in a struct literal of type {tstr(self.w.dealias(t))}")
                for (kk, v), (fn, ft) in zip(e.elts, ut[1]):
                    self.need_assignable(v, self.elt(v, scope, ft), ft, f"field {fn}")
            return t
        if ut[0] in ("slice", "array"):
            et = ut[1] if ut[0] == "slice" else ut[2]
            if ut[0] == "array" and len(e.elts) > ut[1]:
                self.err(e, "too many elements in an array literal")
            for kk, v in e.elts:
                if kk is not None:
                    self.value(kk, scope)
                self.need_assignable(v, self.elt(v, scope, et), et, "slice / array element")
            return t
        if ut[0] == "map":
            for kk, v in e.elts:
                if kk is None:
                    self.err(e, "missing key in a map literal")
                    continue
                self.need_assignable(kk, self.elt(kk, scope, ut[1]), ut[1], "map key")
                self.need_assignable(v, self.elt(v, scope, ut[2]), ut[2], "map value")
            return t
        if ut != T_UNKNOWN:
            self.err(e, f"invalid composite literal type {tstr(t)}")
        return t

    def elt(self, v, scope, et):
        if v.kind == "complit" and v.type is None:
            return self.complit(v, scope, et)
        return self.value(v, scope)

    def call(self, e, scope):
        ft = self.expr(e.fun, scope)
        if ft[0] == "builtin":
            return self.builtin(e, ft[1], scope)
        if ft[0] == "type":                                       # conversion T(x)
            T = ft[1]
            if len(e.args) != 1 or e.ellipsis:
                self.err(e, f"conversion to {tstr(T)} takes exactly one argument")
                return T
            v = self.value(e.args[0], scope)
            if not self.convertible(v, T):
                self.err(e, f"cannot convert {tstr(self.w.dealias(v))} to {tstr(self.w.dealias(T))}")
            return T
        if ft == T_UNKNOWN:
            for a in e.args:
                self.expr(a, scope)
            return T_UNKNOWN
        if ft[0] != "func" and self.w.underlying(ft)[0] == "func":        # a value of a named function type
            ft = self.w.underlying(ft)
        if ft[0] != "func":
            self.err(e, f"cannot call a non-function of type {tstr(ft)}")
            for a in e.args:
                self.expr(a, scope)
            return T_UNKNOWN
        params, results, variadic = ft[1], ft[2], ft[3]
        args = [self.value(a, scope) for a in e.args]
        if len(args) == 1 and args[0][0] == "tuple":
            args = list(args[0][1])
        what = self.callee_name(e.fun)
        if variadic:
            fixed = params[:-1]
            if e.ellipsis:
                if len(args) != len(params):
                    self.err(e, f"{what}: wrong argument count with ...")
                else:
                    for a, p, n in zip(args, fixed, e.args):
                        self.need_assignable(n, a, p, f"argument of {what}")
                    self.need_assignable(e.args[-1], args[-1], ("slice", params[-1]), f"variadic argument of {what}")
            else:
                if len(args) < len(fixed):
                    self.err(e, f"{what}: not enough arguments ({len(args)} for at least {len(fixed)})")
                for i, (a, n) in enumerate(zip(args, e.args)):
                    p = fixed[i] if i < len(fixed) else params[-1]
                    self.need_assignable(n, a, p, f"argument {i + 1} of {what}")
        else:
            if e.ellipsis:
                self.err(e, f"{what} is not variadic")
            if len(args) != len(params):
                self.err(e, f"{what}: {len(args)} argument(s) for {len(params)} parameter(s)")
            else:
                for i, (a, p, n) in enumerate(zip(args, params, e.args if len(e.args) == len(args) else [e] * len(args))):
                    self.need_assignable(n, a, p, f"argument {i + 1} of {what}")
        if len(results) == 0:
            return ("tuple", ())
        if len(results) == 1:
            return results[0]
        return ("tuple", tuple(results))

    @staticmethod
    def callee_name(f):
        if f.kind == "ident":
            return f.name
        if f.kind == "selector":
            base = f.x.name if f.x.kind == "ident" else "…"
            return f"{base}.{f.sel}"
        return "function value"

    def convertible(self, v, T):
        if v == T_UNKNOWN or T == T_UNKNOWN:
            return True
        if self.w.assignable(v, T):
            return True
        uv, uT = self.w.underlying(v) if v[0] != "untyped" else v, self.w.underlying(T)
        Td = self.w.dealias(T)
        vd = self.w.dealias(v) if v[0] != "untyped" else v
        if uv == uT:
            return True
        # unsafe.Pointer <-> any pointer / uintptr
        if Td == T_UPTR:
            return vd[0] == "ptr" or uv == T_basic("uintptr") or v == T_NIL
        if vd == T_UPTR:
            return Td[0] == "ptr" or uT == T_basic("uintptr")
        if uT[0] == "ptr" and uv[0] == "ptr" and self.w.underlying(uT[1]) == self.w.underlying(uv[1]) and Td[0] != "named" and vd[0] != "named":
            return True
        num = lambda t: t[0] == "basic" and (t[1] in INTS or t[1] in FLOATS)      # noqa: E731
        if uT[0] == "basic":
            if v[0] == "untyped":
                if v[1] in ("int", "rune"):
                    return uT[1] in INTS or uT[1] in FLOATS or uT[1] == "string"
                if v[1] == "float":
                    return uT[1] in FLOATS or uT[1] in INTS          # only if representable; the shim uses none
                return False
            if num(uT) and num(uv):
                return True
            if uT == T_STRING and (uv == ("slice", T_basic("uint8")) or num(uv)):
                return True
        if uT == ("slice", T_basic("uint8")) and uv == T_STRING:
            return True
        return False

    def builtin(self, e, name, scope):
        a = e.args
        if name in ("len", "cap"):
            if len(a) != 1:
                self.err(e, f"{name} takes one argument")
                return T_INT
            t = self.w.underlying(self.value(a[0], scope))
            if t[0] not in ("slice", "array", "map", "unknown") and t != T_STRING and not (t[0] == "ptr" and self.w.underlying(t[1])[0] == "array"):
                self.err(e, f"invalid argument of type {tstr(t)} for {name}")
            return T_INT
        if name == "make":
            if not a:
                self.err(e, "make needs a type")
                return T_UNKNOWN
            tt = self.expr(a[0], scope)
            if tt[0] != "type":
                self.err(e, "first argument of make is not a type")
                return T_UNKNOWN
            u = self.w.underlying(tt[1])
            if u[0] not in ("slice", "map"):
                self.err(e, f"cannot make {tstr(tt[1])}")
            if u[0] == "slice" and len(a) not in (2, 3):
                self.err(e, "make([]T) needs a length")
            for x in a[1:]:
                if not self.is_integer(self.value(x, scope)):
                    self.err(e, "make: size is not an integer")
            return tt[1]
        if name == "new":
            tt = self.expr(a[0], scope) if len(a) == 1 else T_UNKNOWN
            if tt[0] != "type":
                self.err(e, "new needs a type")
                return T_UNKNOWN
            return ("ptr", tt[1])
        if name == "append":
            if not a:
                self.err(e, "append needs a slice")
                return T_UNKNOWN
            st = self.value(a[0], scope)
            us = self.w.underlying(st)
            if us[0] != "slice":
                if st != T_UNKNOWN:
                    self.err(e, f"first argument of append is {tstr(st)}, not a slice")
                for x in a[1:]:
                    self.value(x, scope)
                return st
            if e.ellipsis:
                if len(a) != 2:
                    self.err(e, "append(s, x...) takes exactly two arguments")
                else:
                    xt = self.value(a[1], scope)
                    ux = self.w.underlying(xt)
                    ok = ux[0] == "slice" and self.w.identical(ux[1], us[1])
                    if not ok and us[1] == T_basic("uint8") and ux == T_STRING:
                        ok = True
                    if not ok and xt != T_UNKNOWN:
                        self.err(e, f"cannot use {tstr(self.w.dealias(xt))} as {tstr(('slice', self.w.dealias(us[1])))} in append (element types "
                                    f"{tstr(self.w.dealias(ux[1]) if ux[0] == 'slice' else ux)} and {tstr(self.w.dealias(us[1]))} are different types)")
            else:
                for x in a[1:]:
                    self.need_assignable(x, self.value(x, scope), us[1], "append")
            return st
        if name == "copy":
            if len(a) != 2:
                self.err(e, "copy takes two arguments")
                return T_INT
            d, s = self.value(a[0], scope), self.value(a[1], scope)
            ud, us = self.w.underlying(d), self.w.underlying(s)
            if ud[0] != "slice" and ud != T_UNKNOWN:
                self.err(e, f"copy: destination is {tstr(d)}, not a slice")
            elif us[0] == "slice" and ud[0] == "slice" and not self.w.identical(ud[1], us[1]):
                self.err(e, f"copy: element types differ: {tstr(self.w.dealias(d))} and {tstr(self.w.dealias(s))}")
            elif us[0] != "slice" and us != T_UNKNOWN and not (us == T_STRING and ud == ("slice", T_basic("uint8"))):
                self.err(e, f"copy: source is {tstr(s)}, not a slice")
            return T_INT
        if name == "panic":
            if len(a) != 1:
                self.err(e, "panic takes one argument")
            else:
                self.value(a[0], scope)
            return ("tuple", ())
        if name == "recover":
            return T_IFACE
        if name == "delete":
            if len(a) != 2:
                self.err(e, "delete takes two arguments")
            else:
                m = self.w.underlying(self.value(a[0], scope))
                kt = self.value(a[1], scope)
                if m[0] != "map":
                    self.err(e, "delete: not a map")
                else:
                    self.need_assignable(a[1], kt, m[1], "delete key")
            return ("tuple", ())
        for x in a:
            self.value(x, scope)
        return ("tuple", ())


# --------------------------------------------------------------------------------------------------------- driver

def read_dir(d, tests=False):
    out = []
    for f in sorted(os.listdir(d)):
        if f.endswith(".go") and (tests or not f.endswith("_test.go")):
            with open(os.path.join(d, f)) as fh:
                out.append((os.path.join(d, f), fh.read()))
    return out


def build_world(reference_root, header_path, ref_module="github.com/thedonutfactory/go-tfhe", lenient_std=False):
    """Reference packages (top-level declarations only) + C + std."""
    w = World(lenient_std)
    with open(header_path) as fh:
        w.load_c_header(fh.read())
    ref = []
    for d in sorted(os.listdir(reference_root)):
        full = os.path.join(reference_root, d)
        if os.path.isdir(full) and any(f.endswith(".go") and not f.endswith("_test.go") for f in os.listdir(full)):
            pkg, _ = w.load_package(f"{ref_module}/{d}", read_dir(full), name_hint=d, bodies=False)
            ref.append(pkg)
    return w, ref


def check_shim(reference_root, header_path, shim_root, shim_module="github.com/thedonutfactory/go-tfhe-gpu", extra_sources=None, lenient_std=False):
    """Returns (errors, stats).  extra_sources: {import path: [(fname, src)]} checked as additional packages (tests feed
    known-bad sources through this)."""
    w, ref = build_world(reference_root, header_path, lenient_std=lenient_std)
    shim = []
    if shim_root:
        for d in sorted(os.listdir(shim_root)):
            full = os.path.join(shim_root, d)
            if os.path.isdir(full) and any(f.endswith(".go") for f in os.listdir(full)):
                pkg, asts = w.load_package(f"{shim_module}/{d}", read_dir(full, tests=True), name_hint=d, bodies=True)
                shim.append((pkg, asts))
    for path, files in (extra_sources or {}).items():
        pkg, asts = w.load_package(path, files, name_hint=path.rsplit("/", 1)[-1], bodies=True)
        shim.append((pkg, asts))
    # types of every package must exist before any signature is resolved
    for p in ref:
        w.resolve_package(p, strict=False)
    for p, _ in shim:
        w.resolve_package(p)
    errors = list(w.errors)
    stats = {"files": 0, "funcs": 0, "reference_packages": len(ref)}
    for p, asts in shim:
        # package-level vars with inferred types first (their initialisers may reference functions of the package)
        for a in asts:
            stats["files"] += 1
            stats["funcs"] += sum(1 for d in a.decls if d.kind == "funcdecl")
            if a.package.endswith("_test"):
                # an external test package: its own (empty) declarations, same directory
                tp = Package(a.package, p.path + "_test")
                tp._declared_types, tp._asts = {d.name for d in a.decls if d.kind == "typedecl"}, [a]
                w.by_path[tp.path] = tp
                saved = a.package
                a.package = "x"                     # resolve_package skips *_test packages; this one is the package itself
                w.resolve_package(tp)
                a.package = saved
                errors += w.errors[len(errors):] if False else []
                errors += Checker(w, tp, a).check()
            else:
                errors += Checker(w, p, a).check()
    seen, out = set(), []
    for e in errors + [x for x in w.errors if x not in errors]:
        if e not in seen:
            seen.add(e)
            out.append(e)
    return out, stats


if __name__ == "__main__":
    import sys
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    errs, st = check_shim(sys.argv[1] if len(sys.argv) > 1 else "/root/reference", os.path.join(root, "include", "tfhe_hip.h"),
                          os.path.join(root, "shim", "go"))
    for e in errs:
        print(e)
    print(f"{len(errs)} error(s); {st}")
    sys.exit(1 if errs else 0)
