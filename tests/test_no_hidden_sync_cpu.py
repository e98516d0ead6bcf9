"""nothing reachable from vlsat_forward / vlsat_plan_create / vlsat_plan_destroy may wait for the whole
device or issue a blocking copy.  Static check of the two translation units that hold them (the GPU-side evidence is
the rocprofv3 HIP-API trace under profiles/)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cvpr2023-vlsat_amd", "csrc")
BLOCKING = re.compile(r"\b(hipDeviceSynchronize|hipStreamSynchronize|hipMemcpy|hipMemcpy2D|hipMemset|hipMalloc|hipFree|"
                      r"hipHostMalloc|hipEventSynchronize)\s*\(")


def _code(path):
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return "\n".join(l.split("//")[0] for l in txt.splitlines())


def _function_body(code, name):
    i = code.index(name + "(")
    j = code.index("{", i)
    depth, k = 0, j
    while True:
        depth += code[k] == "{"
        depth -= code[k] == "}"
        if depth == 0:
            return code[j:k]
        k += 1


def test_forward_has_no_device_wide_wait_or_blocking_copy():
    code = _code(os.path.join(CSRC, "engine_forward.hip"))
    body = code[:code.index("int vlsat_profile_enable")]          # everything the forward can reach (profile_read may wait)
    hits = BLOCKING.findall(body)
    assert not hits, f"blocking HIP calls reachable from vlsat_forward: {hits}"


def test_plan_create_and_destroy_do_not_wait_for_the_device():
    code = _code(os.path.join(CSRC, "engine_plan.hip"))
    for fn in ("vlsat_plan_destroy", "sweep_trash", "take_event", "give_event"):
        hits = [h for h in BLOCKING.findall(_function_body(code, fn)) if h != "hipFree"]     # hipFree only behind a completed event
        assert not hits, (fn, hits)
    create = _function_body(code, "vlsat_plan_create")
    hits = [h for h in BLOCKING.findall(create) if h not in ("hipMalloc",)]                  # a NEW arena is allocated; recycled ones are not
    assert not hits, hits
    assert "hipMemcpyAsync" in create and "hipStreamWaitEvent" in create
    # the only host-side wait on this path: all 16 pinned staging buffers still in flight
    staging = _function_body(code, "take_staging")
    assert BLOCKING.findall(staging).count("hipEventSynchronize") == 1
