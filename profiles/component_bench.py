"""Isolated nearest-hit loop / shading block, cycles per wave per repetition (s_memtime)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracingmin_amd as rtm
L = rtm.lib()
fn = C.CDLL(rtm._lib.LIB_PATH).rtm_debug_component_bench
fn.restype = C.c_int
fn.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
data = rtm.LoadData(os.path.join(os.path.dirname(__file__), "..", "scenes", "cornellBoxSetting.json")).data
st, arr, n = data.to_c()
names = {0: "nearest ref loop", 1: "nearest fast loop", 2: "nearest fast chunk8", 10: "shade ref", 11: "shade spec"}
for pad, label in ((0, "max occupancy"), (9000, "4 waves/SIMD"), (39000, "1 wave/SIMD (3/CU)")):
    for which in (0, 1, 2, 10, 11):
        c = C.c_double()
        rc = fn(which, arr, n, 2000, 256 * 32, pad, C.byref(c))
        print(f"{label:22s} {names[which]:22s} rc={rc} {c.value:9.1f} cycles/rep/wave", flush=True)
