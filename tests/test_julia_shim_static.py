"""The Julia shim (rome.jl_amd/julia/RoMEMI355Ext.jl) cannot be executed in this image (no Julia).  What CAN be checked without running it is
the part where a mistake corrupts memory silently: the `struct`s it passes by reference must have the fields of include/rome_mi355.h in
the same order with the same widths, its positional constructors must pass one value per field, and every `ccall` must name an
exported symbol with the prototype's argument count and argument classes."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "rome.jl_amd", "julia", "RoMEMI355Ext.jl")
HDR = os.path.join(ROOT, "include", "rome_mi355.h")


def _c_source():
    s = open(HDR).read()
    s = re.sub(r"/\*.*?\*/", " ", s, flags=re.S)
    return re.sub(r"//[^\n]*", " ", s)


def _c_class(decl):
    """'const double*' -> Ptr{Float64} etc.  Opaque handles and void* are Ptr{Cvoid}; pointers to structs are passed as Ref{...}."""
    d = decl.replace("const", " ").strip()
    stars = d.count("*")
    base = d.replace("*", " ").split()[0]
    scalar = {"int32_t": "Int32", "int": "Int32", "uint32_t": "UInt32", "int64_t": "Int64", "uint64_t": "UInt64", "double": "Float64"}
    if stars == 0:
        return scalar.get(base, base)
    if base in scalar and stars == 1:
        return "Ptr{%s}" % scalar[base]
    return "Ptr"     # handles, handle out-parameters, struct pointers


def c_struct(name):
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), _c_source(), flags=re.S).group(1)
    fields = []
    for stmt in body.split(";"):
        stmt = " ".join(stmt.split())
        if not stmt:
            continue
        m = re.match(r"((?:const )?\w+\s*\**)\s*(.*)", stmt)
        typ, names = m.group(1), m.group(2)
        for n in names.split(","):
            n = n.strip()
            ptr = n.startswith("*")
            fields.append((n.lstrip("* "), _c_class(typ + ("*" if ptr else ""))))
    return fields


def jl_source():
    return "\n".join(re.sub(r"#.*", "", re.sub(r"#=.*?=#", "", ln)) for ln in open(JL).read().split("\n"))


def jl_struct(name):
    body = re.search(r"^struct %s\b(.*?)^end" % name, jl_source(), flags=re.S | re.M).group(1)
    out = []
    for stmt in re.split(r"[;\n]", body):
        stmt = stmt.strip()
        if stmt:
            n, t = stmt.split("::")
            out.append((n.strip(), t.strip()))
    return out


JL_TO_CLASS = {"Int32": "Int32", "Cint": "Int32", "UInt32": "UInt32", "Int64": "Int64", "UInt64": "UInt64", "Float64": "Float64",
               "Ptr{Float64}": "Ptr{Float64}", "Ptr{Int32}": "Ptr{Int32}", "RomeCliqueHost": "rome_clique_host"}


def _same_fields(jl_name, c_name):
    jf, cf = jl_struct(jl_name), c_struct(c_name)
    assert [n for n, _ in jf] == [n for n, _ in cf], (jl_name, "field names / order differ from " + c_name)
    for (n, jt), (_, ct) in zip(jf, cf):
        assert JL_TO_CLASS[jt] == ct, "%s.%s: Julia %s vs C %s" % (jl_name, n, jt, ct)
    return len(jf)


def test_structs_mirror_the_header_field_by_field():
    assert _same_fields("RomeOpts", "rome_opts") == 12
    assert _same_fields("RomeCliqueHost", "rome_clique_host") > 40
    assert _same_fields("RomeCliqueUpsolveHost", "rome_clique_upsolve_host") > 25


def _call_args(src, start):
    """the top-level comma-separated arguments of the call whose '(' is at src[start]"""
    depth, args, cur, i = 0, [], "", start
    while True:
        ch = src[i]
        if ch in "([{":
            depth += 1
            if depth > 1:
                cur += ch
        elif ch in ")]}":
            depth -= 1
            if depth == 0:
                if cur.strip():
                    args.append(cur.strip())
                return args
            cur += ch
        elif ch == "," and depth == 1:
            args.append(cur.strip()); cur = ""
        else:
            cur += ch
        i += 1


def test_positional_constructors_pass_one_value_per_field():
    src = jl_source()
    n_fields = {n: len(jl_struct(n)) for n in ("RomeOpts", "RomeCliqueHost", "RomeCliqueUpsolveHost")}
    seen = {n: 0 for n in n_fields}
    for name, want in n_fields.items():
        for m in re.finditer(r"(?<![\w{])%s\(" % name, src):
            if src[:m.start()].rstrip().endswith("struct"):
                continue
            args = _call_args(src, m.end() - 1)
            assert len(args) == want, "%s(...) at offset %d passes %d values for %d fields" % (name, m.start(), len(args), want)
            seen[name] += 1
    assert seen["RomeOpts"] >= 4 and seen["RomeCliqueHost"] == 1 and seen["RomeCliqueUpsolveHost"] == 2, seen   # (one-shot entry + plan)


def c_prototypes():
    protos = {}
    for m in re.finditer(r"(?:^|\n)\s*(?:int|void|const char\s*\*)\s+(rome_\w+)\s*\(([^;{]*?)\)\s*;", _c_source()):
        params = " ".join(m.group(2).split())
        plist = [] if params in ("", "void") else [p.strip() for p in params.split(",")]
        cls = []
        for p in plist:
            # drop the parameter name if there is one: the class is in the type words and stars
            toks = p.replace("*", " * ").split()
            if len(toks) > 1 and toks[-1] != "*" and toks[-1] not in ("int32_t", "double", "int", "uint32_t", "uint64_t", "int64_t", "rome_ctx", "rome_opts"):
                toks = toks[:-1]
            cls.append(_c_class(" ".join(toks)))
        protos[m.group(1)] = cls
    return protos


def test_every_ccall_matches_its_prototype():
    src, protos = jl_source(), c_prototypes()
    assert len(protos) >= 60
    checked = 0
    for m in re.finditer(r"ccall\(", src):
        args = _call_args(src, m.end() - 1)
        sym = re.match(r"\(\s*:(\w+)\s*,\s*LIB\s*\)", args[0])
        if not sym:                       # (sym, LIB) with a computed symbol: sample_prior, checked below
            continue
        name = sym.group(1)
        assert name in protos, "ccall of %s: not declared in include/rome_mi355.h" % name
        jl_types = _call_args(args[2] + " ", 0) if args[2].startswith("(") else None
        assert jl_types is not None, name
        jl_types = [t for t in jl_types if t]
        want = protos[name]
        assert len(jl_types) == len(want), "%s: %d Julia argument types for %d C parameters" % (name, len(jl_types), len(want))
        assert len(args) - 3 == len(want), "%s: %d values passed for %d parameters" % (name, len(args) - 3, len(want))
        for k, (jt, ct) in enumerate(zip(jl_types, want)):
            if ct == "Ptr":
                ok = jt.startswith("Ptr{") or jt.startswith("Ref{")
            elif ct.startswith("Ptr{"):
                ok = jt == ct
            else:
                ok = JL_TO_CLASS.get(jt) == ct
            assert ok, "%s argument %d: Julia %s vs C %s" % (name, k, jt, ct)
        checked += 1
    assert checked >= 22
    for name in ("rome_sample_priorpose2", "rome_sample_priorpose3"):      # the computed-symbol ccall of sample_prior: same 7-parameter shape
        assert protos[name] == ["Ptr", "Ptr", "Int32", "Ptr{Float64}", "Ptr{Float64}", "Ptr{Float64}", "Ptr{Float64}"], protos[name]
        assert ":" + name in src
