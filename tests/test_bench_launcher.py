"""not-gpu: `python bench.py --gpus N` with no launcher environment starts its N ranks itself (the form the driver uses for
the N = 1, 2, 4, 8 scaling runs), relays ONE JSON line and hands a failing rank's exit code on.  The ranks run the stub step of
bench.py (gloo, CPU, no device): this covers the launch / rendezvous / timing-protocol plumbing, not the kernels.  The parent
must not touch the GPU (a process that initialised HIP may not start GPU children on the pool): checked on its source."""
import ast
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(extra, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + extra, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_gpus_2_without_a_launcher_starts_two_ranks_and_prints_one_line():
    r = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--stub-step"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 4 and rec["warmup"] == 1 and rec["config"]["parallelism"] == "dp2"
    assert rec["scaling"] == "weak" and rec["value"] > 0


def test_a_failing_rank_makes_the_parent_exit_non_zero_without_a_line():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "0", "--stub-step"], {"UDA_CLR_STUB_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], r.stdout


def test_under_a_launcher_the_rank_count_must_match():
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "0", "--stub-step"],
             {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "disagree" in (r.stderr + r.stdout)


def test_single_rank_form_is_unchanged():
    r = _run(["--gpus", "1", "--steps", "2", "--warmup", "0", "--stub-step"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip())["n_gpus"] == 1


def test_the_launching_parent_makes_no_device_call():
    """launch_ranks() may only wait on its child: no torch.cuda / HIP / kernel-library use anywhere in its body, and main()
    reaches it before any device call."""
    src = open(BENCH).read()
    tree = ast.parse(src)
    fn = {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef)}
    node = fn["launch_ranks"]
    code = "\n".join(ast.unparse(st) for st in node.body[1:])            # body without the docstring
    assert ast.get_docstring(node)
    for word in ("cuda", "load_library", "uda_clr_amd", "hip", "set_device", "init_process_group"):
        assert word not in code, word
    main_src = ast.get_source_segment(src, fn["main"])
    assert main_src.index("launch_ranks(") < main_src.index("torch.cuda")
