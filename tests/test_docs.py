"""Documentation drift: every test the documents cite exists, every file path they cite exists."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["DESIGN.md", "NOTEBOOK.md", "README.md", "INTEGRATION.md", "profiles/README.md", "scripts/README.md"]


def _defined_tests():
    names = set()
    for f in glob.glob(os.path.join(ROOT, "tests", "**", "test_*.py"), recursive=True):
        names.update(re.findall(r"^def (test_\w+)", open(f).read(), flags=re.M))
    return names


def test_every_cited_test_exists():
    defined = _defined_tests()
    missing = {}
    for doc in DOCS:
        path = os.path.join(ROOT, doc)
        if not os.path.isfile(path):
            continue
        for name in set(re.findall(r"\btest_[a-z0-9_]+\b", open(path).read())):
            if name.endswith("_"):      # a prefix such as test_rowquad_*
                ok = any(d.startswith(name) for d in defined)
            else:
                ok = name in defined or os.path.isfile(os.path.join(ROOT, "tests", name + ".py"))
            if not ok:
                missing.setdefault(doc, []).append(name)
    assert not missing, missing


def test_every_cited_source_file_exists():
    missing = {}
    pat = re.compile(r"(?<![\w/-])((?:docs/archive/scripts|docs/archive/experiments|planedepth_amd|tests|scripts|oracle|include|profiles)/[\w./-]+\.(?:py|hip\.txt|hip|h|sh|md|json|npz|csv))(?![\w.])")
    for doc in DOCS:
        path = os.path.join(ROOT, doc)
        if not os.path.isfile(path):
            continue
        for rel in set(pat.findall(open(path).read())):
            if not os.path.exists(os.path.join(ROOT, rel)):
                missing.setdefault(doc, []).append(rel)
    assert not missing, missing
