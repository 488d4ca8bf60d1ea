"""Every entry point of include/kbnet_hip.h called with null pointers and zero sizes: each must answer with an error status (sizes: 0), none may
take the process down.  Run by tests/test_host_cpu.py in a process of its own (no GPU needed: arguments are checked before anything is launched)."""
import sys, ctypes as C
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kbnet_amd as kb
lib = kb._lib.load()
bad = []
for name, (res, args) in sorted(kb._lib.SIGNATURES.items()):
    if name in ("kbn_version", "kbn_status_string", "kbn_reload_env", "kbn_get_autotune", "kbn_set_autotune", "kbn_knob"):
        continue
    zero = []
    for a in args:
        if a in (C.c_void_p, C.c_char_p) or (hasattr(a, "_type_") and not isinstance(a._type_, str)):
            zero.append(None)
        elif a in (C.c_float, C.c_double):
            zero.append(0.0)
        else:
            zero.append(0)
    print(name, flush=True)
    rc = getattr(lib, name)(*zero)
    if res is C.c_int and rc == 0:
        bad.append(name)
    if res is C.c_size_t and rc != 0:
        bad.append(name + " (size)")
print("accepted all-zero arguments:", bad)
