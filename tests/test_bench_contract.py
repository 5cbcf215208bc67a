"""bench.py's contract, CPU side: the `roofline` kernel is chosen by measurement (a survey of every own MFMA kernel), never hard-coded,
and the tracked evidence agrees with itself -- the kernel named on the committed bench line is the first own kernel of the committed
rocprofv3 summary of the same command."""
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_source():
    with open(os.path.join(ROOT, "bench.py")) as f:
        return f.read()


def test_roofline_kernel_is_not_hard_coded():
    src = _bench_source()
    assert "survey_kernels(eager_step" in src and "survey_kernels(step" in src          # headline and supplementary points
    # no default instance left: --timer-instance is an override only
    m = re.search(r'add_argument\("--timer-instance", default=(\w+)', src)
    assert m and m.group(1) == "None"
    assert 'timer_instance="' not in src


def test_every_timed_tag_has_a_symbol():
    """every `timed("tag", ...)` call site in rel_pose_amd/ops.py maps to a kernel symbol in bench.TAG_SYMBOLS (a survey winner without one
    would put a bare tag, not a kernel name, on the bench line)"""
    with open(os.path.join(ROOT, "rel_pose_amd", "ops.py")) as f:
        ops_src = f.read()
    tags = set()
    for m in re.finditer(r'timed\("([a-z0-9_]+)"((?:\s*\+\s*\(.*?\))*)', ops_src):
        base, rest = m.group(1), m.group(2)
        variants = [base]
        for suf in re.findall(r'"(_[a-z0-9]+)" if', rest):
            variants = variants + [v + suf for v in variants]
        tags.update(variants)
    src = _bench_source()
    table = dict(re.findall(r'"([a-z0-9_]+)":\s*"([^"]+)"', src[src.index("TAG_SYMBOLS = {"):src.index("def tag_symbol")]))
    conditional = set()
    for l in ops_src.splitlines():          # `timed("a" if c else "b", ...)` call sites
        m = re.search(r'timed\("([a-z0-9_]+)" if .*? else "([a-z0-9_]+)"', l)
        if m:
            conditional.update(m.groups())
    missing = sorted(t for t in (tags | conditional) if t not in table)
    assert not missing, "ops.timed tags without a kernel symbol in bench.TAG_SYMBOLS: %s" % missing


def _latest(pattern):
    d = os.path.join(ROOT, "profiles")
    names = sorted(n for n in os.listdir(d) if re.fullmatch(pattern, n))
    return os.path.join(d, names[-1]) if names else None


def test_committed_bench_line_names_the_dominant_kernel_of_the_committed_profile():
    bench, summary = _latest(r"r6_bench\.json"), _latest(r"r6_full_step_summary\.txt")
    if not (bench and summary):
        pytest.skip("round-6 bench line / kernel summary not committed yet")
    with open(bench) as f:
        rec = json.loads(f.readline())
    kernel = rec["roofline"]["kernel"]
    assert rec["roofline"]["kernel_chosen_by"].startswith("survey")
    assert rec["roofline"]["survey"]["top"][0]["kernel"] == kernel
    own = ("_kernel", )
    foreign = ("igemm_", "at::native", "miopen", "Cijk", "SubTensor", "ck::", "naive_conv", "batched_transpose", "hip::")
    first = None
    with open(summary) as f:
        for line in f:
            if "calls=" not in line:
                continue
            name = line.split("/step=")[1].split("ms", 1)[1].strip()
            if any(x in name for x in foreign) or not any(x in name for x in own):
                continue
            first = name
            break
    assert first is not None
    assert kernel.replace("rpgemm::", "") in first.replace("rpgemm::", ""), (kernel, first)
