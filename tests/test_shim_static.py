"""CPU: the Go shim against the C header it binds, statically (there is no Go toolchain in the build image, so the shim has
never met a compiler).  Every `C.bftkv_*(...)` call in shim/**/*.go must name a function include/bftkv_gpu.h declares and pass
as many arguments as the prototype takes; every `C.BFTKV_*` constant must be #defined there; every field the shim sets on a
C struct must be a member of it; braces and parentheses balance; every import of a file is used in it; the C type of every argument
this reading can type (114 of 116) is the prototype's."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "shim")


def go_files():
    out = []
    for dirpath, _, files in os.walk(SHIM):
        out += [os.path.join(dirpath, f) for f in files if f.endswith(".go")]
    assert len(out) >= 7
    return sorted(out)


def strip_go(src):
    """Go source without comments, string / rune literals (replaced by blanks) -- enough for bracket matching and name lookups."""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if src.startswith("//", i):
            j = src.find("\n", i)
            i = n if j < 0 else j
        elif src.startswith("/*", i):
            j = src.find("*/", i + 2)
            assert j >= 0
            out.append("\n" * src.count("\n", i, j))
            i = j + 2
        elif c == '"':
            j = i + 1
            while src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            out.append('""')
            i = j + 1
        elif c == "`":
            j = src.find("`", i + 1)
            out.append('""' + "\n" * src.count("\n", i, j))
            i = j + 1
        elif c == "'":
            j = i + 1
            while src[j] != "'":
                j += 2 if src[j] == "\\" else 1
            out.append("0")
            i = j + 1
        else:
            out.append(c)
            i += 1
    return "".join(out)


def split_args(s):
    """Top-level comma split of an argument list."""
    args, depth, cur = [], 0, []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            args.append("".join(cur).strip())
            cur = []
        else:
            cur.append(ch)
    last = "".join(cur).strip()
    if last:
        args.append(last)
    return args


def call_args(src, open_paren):
    depth = 0
    for j in range(open_paren, len(src)):
        if src[j] == "(":
            depth += 1
        elif src[j] == ")":
            depth -= 1
            if depth == 0:
                return split_args(src[open_paren + 1:j])
    raise AssertionError("unbalanced call")


def header():
    h = open(os.path.join(ROOT, "include", "bftkv_gpu.h")).read() + open(os.path.join(ROOT, "include", "bftkv_host.h")).read()
    h = re.sub(r"/\*.*?\*/", " ", h, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(bftkv_(?:gpu|host)_[a-z_0-9]+)\s*\(([^()]*)\)\s*;", h):
        params = m.group(2).strip()
        protos[m.group(1)] = 0 if params in ("", "void") else len(split_args(params))
    consts = set(re.findall(r"#define\s+(BFTKV_[A-Z_0-9]+)\b", h))
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(bftkv_gpu_[a-z_0-9]+)\s*;", h, flags=re.S):
        fields = set()
        for decl in m.group(1).split(";"):
            decl = decl.strip()
            if decl:
                for part in decl.split(","):
                    fields.add(re.sub(r"\[.*\]", "", part.strip().split()[-1].lstrip("*")))
        structs[m.group(2)] = fields
    return protos, consts, structs


def test_c_calls_match_the_header():
    protos, consts, structs = header()
    assert len(protos) >= 40 and {"bftkv_gpu_pubkey", "bftkv_gpu_qc"} <= set(structs)
    seen = set()
    for path in go_files():
        src = strip_go(open(path).read())
        for m in re.finditer(r"\bC\.(bftkv_(?:gpu|host)_[a-z_0-9]+)\s*\(", src):
            name = m.group(1)
            assert name in protos, "%s: %s is not declared in include/bftkv_gpu.h / bftkv_host.h" % (path, name)
            if name.startswith("bftkv_host_"):
                assert '#include "bftkv_host.h"' in open(path).read(), "%s calls %s without including bftkv_host.h" % (path, name)
            got = len(call_args(src, m.end() - 1))
            assert got == protos[name], "%s: %s called with %d arguments, the prototype takes %d" % (path, name, got, protos[name])
            seen.add(name)
        for name in re.findall(r"\bC\.(BFTKV_[A-Z_0-9]+)\b", src):
            assert name in consts, "%s: C.%s is not #defined in include/bftkv_gpu.h" % (path, name)
        for t in re.findall(r"\bC\.(bftkv_gpu_[a-z_0-9]+)\b(?!\s*\()", src):
            assert t in structs or t in ("bftkv_gpu_ctx", "bftkv_gpu_batcher") or t in protos, (path, t)
    # the seam of the path: every verifying call of the shim is among them
    assert {"bftkv_gpu_init", "bftkv_gpu_keyring_set", "bftkv_gpu_quorum_create", "bftkv_gpu_quorum_destroy",
            "bftkv_gpu_batcher_collective_verify", "bftkv_gpu_batcher_signature_verify", "bftkv_gpu_batcher_cert_verify", "bftkv_gpu_batcher_cert_entity",
            "bftkv_gpu_batcher_message_verify", "bftkv_host_signers_walk", "bftkv_gpu_set_hash_policy",
            # config 5 behind crypto.Threshold (shim/crypto/thresholdgpu)
            "bftkv_gpu_batcher_modmul_product", "bftkv_gpu_batcher_lagrange_combine", "bftkv_gpu_batcher_dsa_calculate_r",
            "bftkv_gpu_batcher_modexp"} <= seen


def test_struct_fields_the_shim_sets_exist():
    _, _, structs = header()
    kr = strip_go(open(os.path.join(SHIM, "crypto", "pgpgpu", "keyring.go")).read())
    for f in re.findall(r"\brec\.([a-z_0-9]+)", kr):
        assert f in structs["bftkv_gpu_pubkey"], f
    q = strip_go(open(os.path.join(SHIM, "crypto", "pgpgpu", "quorum.go")).read())
    for f in re.findall(r"\bqcs\[i\]\.([a-z_0-9]+)", q):
        assert f in structs["bftkv_gpu_qc"], f


def test_go_files_are_structurally_sound():
    for path in go_files():
        raw = open(path).read()
        src = strip_go(raw)
        for a, b in ("{}", "()", "[]"):
            assert src.count(a) == src.count(b), "%s: unbalanced %s%s" % (path, a, b)
        assert re.search(r"^package \w+$", src, flags=re.M), path
        # every import is used (Go refuses an unused one): the package's local name followed by a dot somewhere below
        m = re.search(r"^import \(\n(.*?)^\)", raw, flags=re.S | re.M)
        if not m:
            continue
        body = src[src.index(")", src.index("import (")):]
        for line in m.group(1).splitlines():
            line = line.strip()
            if not line or line.startswith("//"):
                continue
            mm = re.match(r'(?:(\w+)\s+)?"([^"]+)"', line)
            assert mm, (path, line)
            local = mm.group(1) or mm.group(2).rsplit("/", 1)[-1]
            assert re.search(r"\b%s\." % re.escape(local), body), "%s: import %s (%s) is never used" % (path, mm.group(2), local)


def test_interfaces_are_fully_implemented():
    """crypto.Signature / CollectiveSignature / Message / Keyring (crypto/crypto.go:35-70): every method has a receiver in the shim."""
    want = {
        "Signature": ["Verify", "VerifyWithCertificate", "Sign", "Signers", "Issuer", "Certs"],
        "CollectiveSignature": ["Verify", "Sign", "Combine", "Signers"],
        "Message": ["Encrypt", "EncryptStream", "Decrypt"],
        "keyring": ["Register", "Remove", "GetCertById", "GetKeyring"],
    }
    src = "\n".join(strip_go(open(p).read()) for p in go_files() if os.sep + "pgpgpu" + os.sep in p)
    for typ, methods in want.items():
        for meth in methods:
            assert re.search(r"func \(\w+ \*%s\) %s\(" % (typ, meth), src), (typ, meth)


REFERENCE = "/root/reference"
# names of the reference's own API the shim and genvectors call or read (methods of its interfaces, fields of its structs, its
# constants and error values): each must be declared somewhere in the reference tree -- or, for the per-clique accessor, in
# shim/patches.  (x/crypto and the standard library are not checkable here.)
REFERENCE_NAMES = [
    "AddNodes", "Certs", "ChooseQuorum", "Cliques", "Decrypt", "Encrypt", "EncryptStream", "GetCertById", "GetKeyring", "Id", "Instance",
    "IsQuorum", "IsSufficient", "IsThreshold", "Issuer", "Parse", "Register", "Reject", "Remove", "SetSelfNodes", "Sign", "Signers", "Verify",
    "VerifyWithCertificate", "AUTH", "Cert", "Certificate", "CollectiveSignature", "Completed", "Crypto", "Data", "ErrDecryptionFailed",
    "ErrInsufficientNumberOfSignatures", "ErrInvalidSignature", "ErrInvalidTransportSecurityData", "Keyring", "Message", "Signature",
    "SignaturePacket", "SignatureTypeNil", "SignatureTypePGP", "TBS", "TBSS", "Type", "Clique", "Threshold", "Suff", "Min", "Nodes",
    # crypto/thresholdgpu: the hooks of shim/patches/0003 and the reference types they carry
    "CombineHook", "CalculateRHook", "ModExpHook", "Coordinate", "PartialR", "Ri", "Vi",
    # Issuer without ReadEntity on the CPU: the node constructor shim/patches/0004 exports
    "NewNode",
]


def test_reference_api_names_the_shim_uses_exist():
    import pytest
    if not os.path.isdir(REFERENCE):
        pytest.skip("the reference tree is not here")
    decls = []
    for dirpath, _, files in os.walk(REFERENCE):
        decls += [open(os.path.join(dirpath, f), errors="replace").read() for f in files if f.endswith(".go") and not f.endswith("_test.go")]
    for f in sorted(os.listdir(os.path.join(SHIM, "patches"))):
        decls.append("\n".join(ln[1:] for ln in open(os.path.join(SHIM, "patches", f)).read().splitlines() if ln.startswith("+")))
    text = "\n".join(decls)
    used = set()
    for p in go_files():
        src = strip_go(open(p).read())
        used.update(re.findall(r"\.([A-Z]\w*)\b", src))
    for name in REFERENCE_NAMES:
        assert name in used, "%s is not used by the shim any more: drop it from the list" % name
        pat = r"(func (\([^)]*\) )?%s\(|^\s*%s(\(|\s+[\w\[\]\*\.]+|\s*=)|type %s\b|^var %s\b)" % (name, name, name, name)
        assert re.search(pat, text, flags=re.M), "the reference declares no %s" % name


def test_no_variable_is_declared_and_never_used():
    """Go refuses a local variable that is declared and not used.  Rough, but it costs nothing: every name introduced by `:=` or
    `var` inside a function must occur at least once more in that function."""
    for p in go_files():
        src = strip_go(open(p).read())
        for m in re.finditer(r"^func [^\n]*\{\n(.*?)^\}", src, flags=re.S | re.M):
            body = m.group(0)
            names = []
            for d in re.finditer(r"(?:^|[\s;{(])((?:[A-Za-z_]\w*\s*,\s*)*[A-Za-z_]\w*)\s*:=", body):
                names += [x.strip() for x in d.group(1).split(",")]
            for d in re.finditer(r"\bvar\s+((?:[A-Za-z_]\w*\s*,\s*)*[A-Za-z_]\w*)\s", body):
                names += [x.strip() for x in d.group(1).split(",")]
            for name in names:
                if name != "_":
                    assert len(re.findall(r"\b%s\b" % re.escape(name), body)) >= 2, (p, name, body.split("\n")[0])


# ---- cgo is strict about argument types: C.uint32_t where the prototype says uint64_t does not compile ------------------------
def proto_param_types():
    h = open(os.path.join(ROOT, "include", "bftkv_gpu.h")).read() + open(os.path.join(ROOT, "include", "bftkv_host.h")).read()
    h = re.sub(r"/\*.*?\*/", " ", h, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(bftkv_(?:gpu|host)_[a-z_0-9]+)\s*\(([^()]*)\)\s*;", h):
        params = m.group(2).strip()
        types = []
        for p in ([] if params in ("", "void") else split_args(params)):
            p = re.sub(r"\bconst\b", " ", p).strip()
            stars = p.count("*")
            words = p.replace("*", " ").split()
            base = words[0] if len(words) == 1 else " ".join(words[:-1])       # the last word is the parameter's name
            types.append(base.replace("struct ", "") + "*" * stars)
        out[m.group(1)] = types
    return out


def go_c_type(t):
    """'*C.uint8_t' -> 'uint8_t*', 'C.int' -> 'int', 'unsafe.Pointer' -> 'void*'; None for a Go type."""
    t = t.strip()
    if t == "unsafe.Pointer":
        return "void*"
    m = re.fullmatch(r"(\**)C\.(\w+)", t)
    return None if not m else m.group(2) + "*" * len(m.group(1))


def package_facts(paths):
    """Helper functions' return types and struct fields' types of one Go package (only those that are C types)."""
    funcs, fields = {}, {}
    for p in paths:
        src = strip_go(open(p).read())
        for m in re.finditer(r"^func (?:\([^)]*\) )?(\w+)\(([^)]*)\)\s*([^\s{(][^{]*?)?\s*\{", src, flags=re.M):
            ret = go_c_type(m.group(3) or "")
            if ret:
                funcs[m.group(1)] = ret
        for m in re.finditer(r"^type \w+ struct \{\n(.*?)^\}", src, flags=re.S | re.M):
            for ln in m.group(1).split("\n"):
                f = re.match(r"\s*((?:\w+\s*,\s*)*\w+)\s+(\S+)\s*$", ln)
                if f and go_c_type(f.group(2)):
                    for name in f.group(1).split(","):
                        fields.setdefault(name.strip(), set()).add(go_c_type(f.group(2)))
    return funcs, fields


def local_type(fn_src, name):
    """The C type of a local variable or parameter `name` of the function whose text (up to the call) is fn_src; None if unknown."""
    n = re.escape(name)
    for pat, shape in ((r"\bvar\s+(?:\w+\s*,\s*)*%s(?:\s*,\s*\w+)*\s+(\[\w*\])?(\**C\.\w+|unsafe\.Pointer)" % n, "var"),
                       (r"\b%s\s*:=\s*make\(\[\](\**C\.\w+)" % n, "slice"),
                       (r"\b%s\s*:=\s*(C\.\w+)\(" % n, "conv"),
                       (r"\b%s\s*:=\s*\((\*+C\.\w+)\)\(" % n, "conv"),
                       (r"[(,]\s*(?:\w+\s*,\s*)*%s(?:\s*,\s*\w+)*\s+(\**C\.\w+|unsafe\.Pointer)\s*[,)]" % n, "param")):
        m = None
        for m in re.finditer(pat, fn_src):
            pass                                                            # the last declaration before the call
        if m:
            if shape == "var":
                return ("array" if m.group(1) else "value"), go_c_type(m.group(2))
            return ("array" if shape == "slice" else "value"), go_c_type(m.group(1))
    return None


def infer(arg, fn_src, funcs, fields):
    """The C type of one cgo call argument, or None where this rough reading cannot tell."""
    arg = arg.strip()
    m = re.match(r"^(C\.\w+)\(", arg)
    if m and arg.endswith(")"):
        return go_c_type(m.group(1))
    m = re.match(r"^\((\*+C\.\w+)\)\(", arg)
    if m:
        return go_c_type(m.group(1))
    if arg.startswith("unsafe.Pointer("):
        return "void*"
    m = re.match(r"^(\w+)\(", arg)
    if m and arg.endswith(")"):
        return funcs.get(m.group(1))
    m = re.fullmatch(r"&(\w+)(\[0\])?", arg)
    if m:
        lt = local_type(fn_src, m.group(1))
        if lt and (lt[0] == "array") == bool(m.group(2)):
            return lt[1] + "*"
        if lt is None and not m.group(2) and m.group(1) in fields and len(fields[m.group(1)]) == 1:
            return next(iter(fields[m.group(1)])) + "*"
        return None
    m = re.fullmatch(r"&(?:\w+\.)+(\w+)", arg)
    if m and len(fields.get(m.group(1), ())) == 1:
        return next(iter(fields[m.group(1)])) + "*"
    m = re.fullmatch(r"(?:\w+\.)+(\w+)", arg)
    if m and len(fields.get(m.group(1), ())) == 1:
        return next(iter(fields[m.group(1)]))
    if re.fullmatch(r"\w+", arg) and not arg.isdigit() and arg != "nil":
        lt = local_type(fn_src, arg)
        return lt[1] if lt and lt[0] == "value" else None
    return None


INTEGERS = {"int", "uint8_t", "uint16_t", "uint32_t", "uint64_t", "int32_t", "int64_t", "size_t"}


def test_c_call_argument_types_match_the_prototypes():
    """What this reading can type -- conversions C.T(x), casts (*C.T)(p), the package's pointer helpers, &local, &local[0], struct
    fields holding C handles, parameters -- must be the prototype's type exactly (const aside), as cgo demands; integer literals only
    where the prototype takes an integer, nil only where it takes a pointer."""
    protos = proto_param_types()
    by_dir = {}
    for p in go_files():
        by_dir.setdefault(os.path.dirname(p), []).append(p)
    typed = total = 0
    for paths in by_dir.values():
        funcs, fields = package_facts(paths)
        for path in paths:
            src = strip_go(open(path).read())
            for m in re.finditer(r"\bC\.(bftkv_(?:gpu|host)_[a-z_0-9]+)\s*\(", src):
                name = m.group(1)
                start = max(src.rfind("\nfunc ", 0, m.start()), 0)
                fn_src = src[start:m.start()]
                for i, (arg, want) in enumerate(zip(call_args(src, m.end() - 1), protos[name])):
                    total += 1
                    where = "%s: %s argument %d (%s)" % (os.path.relpath(path, ROOT), name, i + 1, arg)
                    if arg.strip().isdigit():
                        assert want in INTEGERS, where + ": an integer literal for " + want
                        typed += 1
                        continue
                    if arg.strip() == "nil":
                        assert want.endswith("*"), where + ": nil for " + want
                        typed += 1
                        continue
                    got = infer(arg, fn_src, funcs, fields)
                    if got is None:
                        continue
                    typed += 1
                    assert got == want, where + ": passes %s, the prototype takes %s" % (got, want)
    assert total >= 100 and typed >= 0.9 * total, (typed, total)
