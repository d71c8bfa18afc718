"""The profile tools' kernel naming (tools/summarize_prof.py): k_spline_scan<ROOT, K, V, PHASE, FAR> is two kernels of one template, told apart by the
PHASE argument -- the last argument is FAR since round 6, which the older rule (the name's last argument) would have read as the phase."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_summarize_prof_names_the_two_scan_kernels_by_phase():
    sp = _load("summarize_prof")
    assert sp.short("void rmi::k_spline_scan<0, unsigned long, 16, 0, 1>(rmi::ScanArgs)") == "k_spline_scan<short>"
    assert sp.short("void rmi::k_spline_scan<3, unsigned int, 32, 0, 2>(rmi::ScanArgs)") == "k_spline_scan<short>"
    assert sp.short("void rmi::k_spline_scan<3, unsigned int, 32, 0, 0>(rmi::ScanArgs)") == "k_spline_scan<short>"
    assert sp.short("void rmi::k_spline_scan<0, double, 16, 1, 0>(rmi::ScanArgs)") == "k_spline_scan<general>"
    assert sp.short("void rmi::k_spline_scan<0, unsigned long, 16, 1") == "k_spline_scan"          # (a truncated name of the stats file: left unnamed)
