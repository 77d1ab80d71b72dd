"""The cgo binding (integration/go) cannot be compiled here (no Go toolchain), so this keeps it in step with the C header by
text: every C function it calls is declared in include/pqdecode.h with that many parameters, every PQD_* constant and every
struct type it names exists there, and every field it reads from a C struct is a member of that struct."""
import os, re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "pqdecode.h")).read()
GO_FILES = [os.path.join(ROOT, "integration", "go", "pqdecode", "pqdecode.go"),
            os.path.join(ROOT, "integration", "go", "layout_gpu", "page_read_gpu.go")]


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def top_level_args(text, open_paren):
    """-> (number of top-level arguments of the call whose '(' is at open_paren, index behind its ')')"""
    depth, args, seen = 0, 0, False
    i = open_paren
    while i < len(text):
        ch = text[i]
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
            if depth == 0:
                return (args + 1 if seen else 0), i + 1
        elif ch == "," and depth == 1:
            args += 1
        elif depth >= 1 and not ch.isspace():
            seen = True
        i += 1
    raise AssertionError("unbalanced call")


def header_functions():
    h = strip_comments(HEADER)
    out = {}
    for m in re.finditer(r"\b(pqd_[a-z0-9_]+)\s*\(", h):
        n, _ = top_level_args(h, m.end() - 1)
        params = h[m.end():h.index(")", m.end())].strip()
        out[m.group(1)] = 0 if params == "void" else n
    return out


def header_structs():
    h = strip_comments(HEADER)
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)?\s*\{(.*?)\}\s*(\w+)\s*;", h, flags=re.S):
        fields = set()
        for decl in m.group(2).split(";"):
            names = re.findall(r"(\w+)\s*(?:\[[^\]]*\])?\s*(?:,|$)", decl.strip())
            fields.update(names)
        out[m.group(3)] = fields
    return out


def test_every_c_call_matches_its_declaration():
    funcs = header_functions()
    assert "pqd_decode_chunk" in funcs and funcs["pqd_last_error"] == 0
    calls = 0
    for path in GO_FILES:
        go = strip_comments(open(path).read())
        for m in re.finditer(r"\bC\.(pqd_[a-z0-9_]+)\(", go):
            name = m.group(1)
            if name not in funcs:      # a type conversion such as C.pqd_page_desc(...) would be a struct, not a function
                assert name in header_structs(), "%s: C.%s is not declared in pqdecode.h" % (os.path.basename(path), name)
                continue
            n, _ = top_level_args(go, m.end() - 1)
            assert n == funcs[name], "%s: C.%s called with %d arguments, declared with %d" % (os.path.basename(path), name, n, funcs[name])
            calls += 1
    assert calls >= 10


def test_every_constant_and_type_exists_in_the_header():
    h = strip_comments(HEADER)
    constants = set(re.findall(r"\b(PQD_[A-Z0-9_]+)\b", h))
    structs = header_structs()
    for path in GO_FILES:
        go = strip_comments(open(path).read())
        for name in set(re.findall(r"\bC\.(PQD_[A-Z0-9_]+)\b", go)):
            assert name in constants, "%s: C.%s is not in pqdecode.h" % (os.path.basename(path), name)
        for name in set(re.findall(r"\bC\.(pqd_[a-z0-9_]+)\b(?!\()", go)):
            assert name in structs or re.search(r"\b%s\b" % name, h), "%s: C.%s is not in pqdecode.h" % (os.path.basename(path), name)


def test_struct_fields_read_by_the_binding_exist():
    structs = header_structs()
    assert {"status", "bad_page"} <= structs["pqd_chunk_result"]
    go = strip_comments(open(GO_FILES[0]).read())
    # variables declared with a C struct type, and the fields read from them
    for var, typ in re.findall(r"\bvar\s+(\w+)\s+C\.(pqd_\w+)\b", go):
        if typ not in structs:
            continue
        for field in set(re.findall(r"\b%s\.(\w+)\b" % re.escape(var), go)):
            go_field = field[1:] if field.startswith("_") else field      # cgo spells a member named like a Go keyword with a leading underscore
            assert go_field in structs[typ], "pqdecode.go reads %s.%s, which %s does not have" % (var, field, typ)
