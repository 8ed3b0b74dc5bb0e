"""Extracts the public signatures of the reference's hot-path functions (AST only; nothing is imported or executed)
into tests/golden/reference_signatures.json.  Run in the build container: python tests/golden/make_signatures.py"""
import ast, json, os, re

REF = "/root/reference/src/squidpy/gr"
HERE = os.path.dirname(os.path.abspath(__file__))
WANT = {"_nhood.py": ["nhood_enrichment", "interaction_matrix"], "_ppatterns.py": ["spatial_autocorr", "co_occurrence"], "_ripley.py": ["ripley"], "_ligrec.py": ["ligrec"],
        "_build.py": ["spatial_neighbors", "spatial_neighbors_knn", "spatial_neighbors_radius", "spatial_neighbors_grid", "spatial_neighbors_delaunay", "spatial_neighbors_from_builder", "mask_graph"]}
out = {}
for fn, names in WANT.items():
    src = open(os.path.join(REF, fn)).read()
    src = re.sub(r"^(class|def) (\w+)\[[^\]]*\]\(", r"\1 \2(", src, flags=re.M)  # PEP 695 generics: py3.10 cannot parse them
    src = re.sub(r"(?m)^type\s+\w+(\[[^\]]*\])?\s*=.*$", "pass", src)  # PEP 695 type aliases
    tree = ast.parse(src)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            a = node.args
            pos = [x.arg for x in a.args]
            defaults = [ast.unparse(d) for d in a.defaults]
            n_no_default = len(pos) - len(defaults)
            out[node.name] = {
                "file": f"src/squidpy/gr/{fn}:{node.lineno}",
                "positional": [{"name": n, "default": (defaults[i - n_no_default] if i >= n_no_default else None)} for i, n in enumerate(pos)],
                "keyword_only": [{"name": k.arg, "default": (ast.unparse(d) if d is not None else None)} for k, d in zip(a.kwonlyargs, a.kw_defaults)],
                "decorators": [ast.unparse(d) for d in node.decorator_list],
            }
json.dump(out, open(os.path.join(HERE, "reference_signatures.json"), "w"), indent=1)
print({k: len(v["positional"]) for k, v in out.items()})
