"""CPU: the documents name only what exists.  Every raisr_hip_* / RNLHandler_* / RNL* identifier that INTEGRATION.md, README.md,
DESIGN.md, docs/ and the FFmpeg files mention is declared in include/ (or is one of the listed non-API names); every
RAISR_HIP_* environment variable they mention is read somewhere in the sources; every repo path they cite exists."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["INTEGRATION.md", "README.md", "DESIGN.md", "docs/CERTIFY.md", "docs/EXPERIMENTS.md", "ffmpeg/README.md", "scripts/README.md", "tools/pin_against_reference/README.md"]


def _read(paths):
    return "\n".join(open(p, errors="replace").read() for p in paths)


def test_api_names_in_the_documents_are_declared():
    headers = _read(glob.glob(os.path.join(ROOT, "include", "*.h")) + glob.glob(os.path.join(ROOT, "include", "raisr", "*.h")))
    declared = set(re.findall(r"\b(raisr_hip_\w+|RNLHandler_\w+|RNL[A-Z]\w+|RAISR_HIP_[A-Z0-9_]+)\b", headers))
    # names that are not C API: kernels / internal helpers / python helpers / file names the documents talk about
    internal = set(re.findall(r"\b(raisr_hip_\w+)\b", _read(glob.glob(os.path.join(ROOT, "video-super-resolution-library_amd", "csrc", "*")) +
                                                            glob.glob(os.path.join(ROOT, "video-super-resolution-library_amd", "*.py")))))
    missing = {}
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        for name in set(re.findall(r"\b(raisr_hip_\w+|RNLHandler_\w+)\b", text)):
            base = name.rstrip("_")
            if name.endswith(("_", ".h", ".py", ".so")) or base in ("raisr_hip_stream", "raisr_hip", "raisr_hip_debug_certify", "raisr_hip_process_host",
                                                                       "raisr_hip_filter_deps"):      # the last one: an FFmpeg configure variable
                continue
            if name in declared or name in internal or any(d.startswith(name) for d in declared):
                continue
            missing.setdefault(doc, []).append(name)
    assert not missing, missing


def test_environment_variables_in_the_documents_are_read_by_the_code():
    # "the code" = what a user of the library runs: the C / HIP sources, the package's python files and bench.py.  A variable that only a
    # script under scripts/ (or a test) mentions is not a knob of the product.  The lab notebook (docs/EXPERIMENTS.md) and scripts/README.md
    # also talk about the switches of rejected experiments, which exist in scripts/*.patch and development builds only: for those two
    # documents the scripts count as well.
    product = _read(glob.glob(os.path.join(ROOT, "video-super-resolution-library_amd", "csrc", "*")) +
                    glob.glob(os.path.join(ROOT, "video-super-resolution-library_amd", "*.py")) + [os.path.join(ROOT, "bench.py")] +
                    glob.glob(os.path.join(ROOT, "include", "*.h")) + glob.glob(os.path.join(ROOT, "include", "raisr", "*.h")) +
                    glob.glob(os.path.join(ROOT, "ffmpeg", "*.c")) + glob.glob(os.path.join(ROOT, "ffmpeg", "*.diff")) +
                    [os.path.join(ROOT, "tools", "pin_against_reference", f) for f in os.listdir(os.path.join(ROOT, "tools", "pin_against_reference"))
                     if f.endswith((".py", ".sh"))])
    lab = product + _read([f for f in glob.glob(os.path.join(ROOT, "scripts", "*")) + glob.glob(os.path.join(ROOT, "scripts", "sessions", "*")) if os.path.isfile(f)] +
                          glob.glob(os.path.join(ROOT, "tests", "*.py")))
    missing = {}
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        srcs = lab if doc in ("docs/EXPERIMENTS.md", "scripts/README.md") else product
        for name in set(re.findall(r"\b(RAISR_[A-Z]+_[A-Z0-9_]+)\b", text)):
            if name.endswith("_") or re.search(r"\b" + re.escape(name) + r"\b", srcs):
                continue
            missing.setdefault(doc, []).append(name)
    assert not missing, missing


def test_repo_paths_cited_in_the_documents_exist():
    missing = {}
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        for path in set(re.findall(r"`((?:scripts|tests|profiles|oracle|include|ffmpeg|docs|video-super-resolution-library_amd)/[\w./\-]+?\.(?:py|sh|c|h|hip|cpp|md|json|diff|patch|txt|csv))`", text)):
            if "*" in path or "<" in path or "{" in path or "…" in path:
                continue
            if path in ("ffmpeg/vf_raisr.c", "ffmpeg/vf_raisr_opencl.c", "ffmpeg/0001-ffmpeg-raisr-filter.patch"):
                continue                                 # files of the REFERENCE tree the documents compare with
            if path == "tests/golden/reference_digests.json":
                continue                                 # written by tools/pin_against_reference/ on a host with Intel IPP; absent here by design
            if not os.path.exists(os.path.join(ROOT, path)):
                missing.setdefault(doc, []).append(path)
    assert not missing, missing


def test_every_script_is_listed_in_scripts_readme():
    text = open(os.path.join(ROOT, "scripts", "README.md")).read()
    names = set(re.findall(r"([\w.{},\-]+\.(?:py|sh|hip|c|patch|txt))\b", text))
    expanded = set()
    for n in names:                                       # `exp_hot_keys_{natural,random}.txt`
        m = re.match(r"(.*)\{([^}]*)\}(.*)", n)
        expanded |= {m.group(1) + alt + m.group(3) for alt in m.group(2).split(",")} if m else {n}
    ranges = [(p, int(a), int(b)) for p, a, b in re.findall(r"`(\w+?)(\d+)\.sh` … `\w+?(\d+)\.sh`", text)]      # `r03_call1.sh` … `r03_call16.sh`
    missing = []
    for fn in sorted(os.listdir(os.path.join(ROOT, "scripts"))):
        if fn == "README.md" or fn in expanded or fn == "sessions":      # sessions/: the GPU sessions' command lists, described as a group
            continue
        m = re.match(r"(\w+?)(\d+)\.sh$", fn)
        if m and any(p == m.group(1) and a <= int(m.group(2)) <= b for p, a, b in ranges):
            continue
        missing.append(fn)
    assert not missing, missing
