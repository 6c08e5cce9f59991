"""CPU: the two C-ABI libraries load and export every symbol include/wsis_hip.h declares."""
import ctypes
import os
import re

import wsis_native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(experimental=None):
    """entry points include/wsis_hip.h declares for the build ``experimental`` says (None: the loaded library's own
    flavour, wsis_experimental()): the declarations inside `#if defined(WSIS_EXPERIMENTAL) && WSIS_EXPERIMENTAL` guards
    belong to the EXPERIMENTAL build only"""
    if experimental is None:
        experimental = wsis_native.experimental()
    text = open(os.path.join(ROOT, "include", "wsis_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    guard = r"#if defined\(WSIS_EXPERIMENTAL\) && WSIS_EXPERIMENTAL\n(.*?)#endif"
    guarded = "".join(re.findall(guard, text, flags=re.S))
    base = re.sub(guard, "", text, flags=re.S)
    names = set(re.findall(r"\b(wsis_\w+)\s*\(", base))
    extra = set(re.findall(r"\b(wsis_\w+)\s*\(", guarded))
    assert extra and not (extra & names)
    return sorted(names | extra) if experimental else sorted(names)


def test_header_symbols_exported():
    names = _declared()
    assert len(names) >= 30
    host = ctypes.CDLL(os.path.join(ROOT, "3d-wsis_amd", "libwsis_host.so"))
    hip = ctypes.CDLL(os.path.join(ROOT, "3d-wsis_amd", "libwsis_hip.so"))
    for n in names:
        lib = host if n.startswith("wsis_host_") else hip
        assert hasattr(lib, n), f"{n} declared in include/wsis_hip.h but not exported"


def test_binding_table_matches_header():
    for flavour in (False, True):
        host_names, hip_names = wsis_native.declared_symbols(experimental=flavour)
        assert sorted(host_names + hip_names) == _declared(flavour)


def test_default_build_does_not_export_the_retired_designs():
    """the default library is what bench.py runs: the entry points of the retired designs exist
    in the EXPERIMENTAL build only, and asking for one of their switches on the default build is an error"""
    import pytest
    hip = ctypes.CDLL(os.path.join(ROOT, "3d-wsis_amd", "libwsis_hip.so"))
    extra = sorted(set(_declared(True)) - set(_declared(False)))
    if wsis_native.experimental():
        assert all(hasattr(hip, n) for n in extra)
        return
    assert not any(hasattr(hip, n) for n in extra)
    for sym in ("wsis_debug_ring_diag", "wsis_debug_deep_phases"):
        assert not hasattr(hip, sym)
    with pytest.raises(wsis_native.WsisError):
        wsis_native.require_experimental("WSIS_DEEP=1")


def test_libraries_load_and_report_version():
    abi = int(re.search(r"#define WSIS_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "wsis_hip.h")).read()).group(1))
    assert wsis_native.host().wsis_host_version() == abi
    assert wsis_native.hip().wsis_version() == abi
    assert wsis_native.hip().wsis_device_count() >= 0


def test_device_ops_refuse_cpu_tensors():
    import pytest
    import torch
    import pointgroup_ops
    import torch_scatter
    with pytest.raises(wsis_native.WsisError):
        pointgroup_ops.voxelization(torch.zeros(4, 3), torch.zeros((2, 3), dtype=torch.int32), 4)
    with pytest.raises(wsis_native.WsisError):
        torch_scatter.scatter(torch.zeros(4, 3), torch.zeros(4, dtype=torch.long), dim=0, reduce="mean")


def test_heads_struct_matches_header():
    """wsis_native.Heads (ctypes) against ``typedef struct wsis_heads`` of include/wsis_hip.h: same fields, same order,
    same array length -- the struct crosses the boundary by pointer"""
    text = open(os.path.join(ROOT, "include", "wsis_hip.h")).read()
    body = re.search(r"typedef struct wsis_heads \{(.*?)\} wsis_heads;", text, flags=re.S).group(1)
    n_max = int(re.search(r"#define WSIS_HEADS_MAX (\d+)", text).group(1))
    assert n_max == wsis_native.HEADS_MAX
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            names.append(re.sub(r"\[.*", "", part.strip().split()[-1].lstrip("*")))
    assert names == [f[0] for f in wsis_native.Heads._fields_]
    assert ctypes.sizeof(wsis_native.Heads) == 4 * 2 + 4 * n_max + 8 * n_max * 17


def test_tuning_knob_list_matches_the_sources_and_warns_on_the_default_build(monkeypatch):
    """every name the C layer reads through tune_env / tune_int / dw2_env (csrc/common.h: live in the EXPERIMENTAL build,
    a compiled-in default otherwise) is in ``unet_native.TUNE_KNOBS``; setting one while the default library is loaded
    warns once instead of being a silent no-op (tools/conv_ab.py, ab_step.py on the wrong flavour would read A == B)"""
    import glob
    import warnings
    from model import unet_native
    names = set()
    for f in glob.glob(os.path.join(ROOT, "3d-wsis_amd", "csrc", "*.h*")):
        names |= set(re.findall(r"(?:tune_env|tune_int|dw2_env|dw3_env)\(\"(WSIS_\w+)\"", open(f).read()))
    assert names and names == set(unet_native.TUNE_KNOBS), sorted(names ^ set(unet_native.TUNE_KNOBS))
    knob = "WSIS_FWD2_WAVES"
    monkeypatch.setenv(knob, "512")
    unet_native._WARNED_KNOBS.discard(knob)
    unet_native._KNOB_SCAN[0] = 0
    with warnings.catch_warnings(record=True) as got:
        warnings.simplefilter("always")
        unet_native._check_experimental_switches()
        unet_native._check_experimental_switches()
    hits = [w for w in got if knob in str(w.message)]
    assert len(hits) == (0 if wsis_native.experimental() else 1)
