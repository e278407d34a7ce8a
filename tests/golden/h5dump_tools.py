"""Helpers for the abundance.h5 goldens (h5dump text): parse datasets and collapse the two constant bias vectors."""
import re


def collapse_constant_bias(dump: str) -> str:
    """Replace the 4096-line DATA blocks of aux/bias_observed and aux/bias_normalized by one line when every value is 1."""
    out, lines, i = [], dump.split("\n"), 0
    in_bias = False
    while i < len(lines):
        ln = lines[i]
        if re.search(r'DATASET "bias_(observed|normalized)"', ln):
            in_bias = True
        if in_bias and ln.strip() == "DATA {":
            j = i + 1
            vals = []
            while lines[j].strip() != "}":
                vals += [x for x in re.sub(r"\(\d+\):", " ", lines[j]).replace(",", " ").split()]
                j += 1
            if vals and all(float(v) == 1.0 for v in vals):
                out += [ln, ln.replace("DATA {", f"(all {len(vals)} values): 1"), lines[j]]
            else:
                out += lines[i:j + 1]
            i = j + 1
            in_bias = False
            continue
        out.append(ln)
        i += 1
    return "\n".join(out)


def parse(dump: str) -> dict:
    """h5dump text -> {dataset path: {"type": str, "dims": int, "chunk": int|None, "deflate": int|None, "strsize": int|None,
    "values": list}} (values: floats / ints as float, strings unquoted; the collapsed bias vectors expand to ones)."""
    res, stack, cur = {}, [], None   # stack of ("group", name) / ("dataset", name) / ("other", None), one per open brace
    lines = dump.split("\n")
    i = 0
    while i < len(lines):
        s = lines[i].strip()
        mg = re.match(r'GROUP "(.*)" \{', s)
        md = re.match(r'DATASET "(.*)" \{', s)
        if mg:
            stack.append(("group", mg.group(1).strip("/")))
        elif md:
            path = "/".join([n for k, n in stack if k == "group" and n] + [md.group(1)])
            cur = res[path] = {"type": None, "dims": None, "chunk": None, "deflate": None, "strsize": None, "values": []}
            stack.append(("dataset", path))
        elif s == "DATA {" and cur is not None:
            j = i + 1
            while lines[j].strip() != "}":
                t = lines[j].strip()
                m2 = re.match(r"\(all (\d+) values\): 1", t)
                if m2:
                    cur["values"] += [1.0] * int(m2.group(1))
                else:
                    t = re.sub(r"^\(\d+\):", "", t).strip().rstrip(",")
                    if t.startswith('"'):
                        cur["values"] += re.findall(r'"([^"]*)"', t)
                    elif t:
                        cur["values"] += [float(x) for x in t.replace(",", " ").split()]
                j += 1
            i = j   # the closing brace of DATA is consumed here
        elif s.endswith("{") and "}" not in s:
            stack.append(("other", None))
        elif s == "}":
            if stack and stack.pop()[0] == "dataset":
                cur = None
        if cur is not None:
            m = re.match(r"DATATYPE\s+(\S+)", s)
            if m:
                cur["type"] = m.group(1)
            m = re.match(r"STRSIZE (\d+);", s)
            if m:
                cur["strsize"] = int(m.group(1))
            m = re.match(r"DATASPACE\s+SIMPLE \{ \( (\d+) \)", s)
            if m:
                cur["dims"] = int(m.group(1))
            m = re.match(r"CHUNKED \( (\d+) \)", s)
            if m:
                cur["chunk"] = int(m.group(1))
            m = re.match(r"COMPRESSION DEFLATE \{ LEVEL (\d+) \}", s)
            if m:
                cur["deflate"] = int(m.group(1))
        i += 1
    return res
