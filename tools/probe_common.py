"""Shared by tools/*.py: make the package run on the probe build of the library (tools/libhippomm_probe.so =
the product sources compiled with -DHMM_PROBE plus tools/csrc/*.hip), so that tuning knobs (hmm_probe_set_*) can be
A/B-ed in one process.  Build it with `python -m hippomm_amd.build --probe`."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def load_probe():
    from hippomm_amd import _lib as L
    path = os.environ.get("HMM_PROBE_LIB") or os.path.join(ROOT, "tools", "libhippomm_probe.so")   # override: A/B of two builds
    if not os.path.exists(path):
        raise SystemExit(f"{path} missing: python -m hippomm_amd.build --probe")
    L._lib = L.bind(path)          # every later _lib.load() in this process returns the probe build
    return L, L._lib


def setter(lib, name):
    fn = getattr(lib, "hmm_probe_set_" + name)
    fn.restype = None
    fn.argtypes = [C.c_int]
    return fn


def event_ms(fn, iters, warmup=2):
    import torch
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def own_power_file():
    """hwmon power1_input of THE GPU this process uses.  A box's sysfs lists every card of the host (eight on these nodes,
    seven of them other people's): match the PCI address of torch's device 0 against /sys/class/drm/card*/device."""
    import glob
    import torch
    pr = torch.cuda.get_device_properties(torch.cuda.current_device())
    want = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
    for dev in glob.glob("/sys/class/drm/card*/device"):
        if os.path.basename(os.path.realpath(dev)).lower() == want:
            files = glob.glob(os.path.join(dev, "hwmon", "hwmon*", "power1_input"))
            if files:
                return files[0]
    return None
