"""Import environment of the golden generators (build container only).

The generators import the REFERENCE from /root/reference and this repository's seeded input recipes
(interpret_quality_amd/synth.py) in one process.  The repository root must NOT be importable there: its `tools/` package
(the import-level drop-in shims `tools.final_common` / `tools.final_util`) is a regular package and would shadow the
reference's namespace package of the same name whatever the order of sys.path.  So `synth` is loaded by file path and
the repository root is removed from sys.path before the reference is imported."""
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def setup():
    """-> the synth module.  Afterwards `import tools...`, `import models...`, `import final_*` resolve to the reference."""
    spec = importlib.util.spec_from_file_location("iq_synth", os.path.join(REPO, "interpret_quality_amd", "synth.py"))
    synth = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(synth)
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or os.getcwd()) != REPO]
    sys.path.insert(0, REF)
    for name in [m for m in sys.modules if m == "tools" or m.startswith("tools.")]:
        del sys.modules[name]
    return synth
