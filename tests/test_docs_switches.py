"""README's list of run-time switches names only switches the library still reads (as a string literal in gnumap_amd/csrc)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sources():
    out = []
    for d, _, fs in os.walk(os.path.join(ROOT, "gnumap_amd", "csrc")):
        for f in fs:
            if f.endswith((".cpp", ".hip", ".h")):
                out.append(os.path.join(d, f))
    return out


def test_readme_switches_exist_in_the_sources():
    readme = open(os.path.join(ROOT, "README.md")).read()
    para = readme[readme.index("Switches (environment"):]
    para = para[: para.index("There is no CPU fallback")]
    names = set(re.findall(r"\bGM_[A-Z0-9_]+\b", para))
    assert len(names) > 20
    text = "\n".join(open(p, errors="replace").read() for p in _sources())
    missing = sorted(n for n in names if f'"{n}"' not in text)          # read through gm_opt("NAME") / getenv("NAME")
    assert not missing, f"README names switches nothing reads: {missing}"
