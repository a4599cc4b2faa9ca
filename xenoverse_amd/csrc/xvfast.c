/* _xvfast — a CPython trampoline for the hottest C-ABI calls of the Python surface (the eager [policy -> step] loop).
 *
 * ctypes marshals every argument of `xv_anymdp_step_info(h, action, obs, ..., mode)` through libffi: ~3 us per call of 11
 * arguments, most of what a 65,536-env step costs the host beside the policy's own torch op.  The entry points of
 * include/xeno.h take pointers and ints only, so a call can be made from Python ints directly:
 *
 *     rc = _xvfast.icall(fn_address, a0, a1, ..., an)        # every a_i a Python int (pointers as ints), at most 14
 *
 * calls `int fn(a0, ..., an)` with exactly n + 1 integer-class arguments (System V x86-64 / AAPCS64: pointers and ints travel in
 * the same 64-bit registers, a callee that declares `int` reads the low half) and returns its int.  No GIL release: only
 * non-blocking launches are routed here (the callers keep ctypes for everything else, and for all of it when this module is
 * not built: a binding shortcut, not a second implementation of anything).
 * The reference has no counterpart: its step() is pure Python (anymdp_env.py:112-132). */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

typedef uint64_t u64;
#define XV_MAX_ARGS 14

static PyObject* xvfast_icall(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
  (void)self;
  if (nargs < 1 || nargs > XV_MAX_ARGS + 1) {
    PyErr_SetString(PyExc_TypeError, "icall(fn_address, *ints): 0 .. 14 integer arguments");
    return NULL;
  }
  u64 a[XV_MAX_ARGS + 1];
  for (Py_ssize_t i = 0; i < nargs; ++i) {
    a[i] = PyLong_AsUnsignedLongLongMask(args[i]);      /* (negative ints wrap: an `int` parameter reads the low 32 bits) */
    if (a[i] == (u64)-1 && PyErr_Occurred()) return NULL;
  }
  if (a[0] == 0) {
    PyErr_SetString(PyExc_ValueError, "icall: null function address");
    return NULL;
  }
  int rc;
  void* f = (void*)(uintptr_t)a[0];
  switch (nargs - 1) {
    case 0: rc = ((int (*)(void))f)(); break;
    case 1: rc = ((int (*)(u64))f)(a[1]); break;
    case 2: rc = ((int (*)(u64, u64))f)(a[1], a[2]); break;
    case 3: rc = ((int (*)(u64, u64, u64))f)(a[1], a[2], a[3]); break;
    case 4: rc = ((int (*)(u64, u64, u64, u64))f)(a[1], a[2], a[3], a[4]); break;
    case 5: rc = ((int (*)(u64, u64, u64, u64, u64))f)(a[1], a[2], a[3], a[4], a[5]); break;
    case 6: rc = ((int (*)(u64, u64, u64, u64, u64, u64))f)(a[1], a[2], a[3], a[4], a[5], a[6]); break;
    case 7: rc = ((int (*)(u64, u64, u64, u64, u64, u64, u64))f)(a[1], a[2], a[3], a[4], a[5], a[6], a[7]); break;
    case 8: rc = ((int (*)(u64, u64, u64, u64, u64, u64, u64, u64))f)(a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8]); break;
    case 9: rc = ((int (*)(u64, u64, u64, u64, u64, u64, u64, u64, u64))f)(a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9]); break;
    case 10: rc = ((int (*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))f)(a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10]); break;
    case 11: rc = ((int (*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))f)(a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11]); break;
    case 12: rc = ((int (*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))f)(a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12]); break;
    case 13: rc = ((int (*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))f)(a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13]); break;
    default: rc = ((int (*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))f)(a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14]); break;
  }
  return PyLong_FromLong((long)rc);
}

static PyMethodDef xvfast_methods[] = {
    {"icall", (PyCFunction)(void (*)(void))xvfast_icall, METH_FASTCALL,
     "icall(fn_address, *ints) -> int: call `int fn(...)` of the C-ABI with integer-class arguments (pointers as ints)"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef xvfast_module = {PyModuleDef_HEAD_INIT, "_xvfast", "trampoline for the hottest xeno.h calls", -1, xvfast_methods,
                                           NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__xvfast(void) { return PyModule_Create(&xvfast_module); }
