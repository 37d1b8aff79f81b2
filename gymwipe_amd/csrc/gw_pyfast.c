/* gw_pyfast.c -- the per-step call from Python without ctypes.
 *
 * env.step() is enqueued about 200 000 times a second; through ctypes each call converts seven Python ints into C arguments
 * by way of generic descriptors (~1 us).  This CPython extension does the same call through METH_FASTCALL (~0.1 us).  It holds
 * no logic: it calls the C-ABI entry points whose addresses the Python side hands it (taken from the loaded
 * libgymwipe_amd.so with ctypes), so it links against nothing but libpython's ABI and cannot drift from the library.
 * Built in-tree by the Makefile next to the library; gymwipe_amd/_native.py falls back to ctypes when it is absent. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

typedef int (*gw_step_fn)(void*, const int32_t*, const int32_t*, int32_t*, float*, uint8_t*, void*);
typedef int (*gw_pend_fn)(void*, void*, const int32_t*, const int32_t*, int32_t*, float*, double*, void*);
typedef int (*gw_step_fb_fn)(void*, const int32_t*, const int32_t*, int32_t*, float*, uint8_t*, uint8_t*, void*);
static gw_step_fn g_step = NULL;
static gw_pend_fn g_pend = NULL;
static gw_step_fb_fn g_step_fb = NULL;

static int as_ptr(PyObject* o, void** out)
{
    const unsigned long long v = PyLong_AsUnsignedLongLong(o);
    if (v == (unsigned long long)-1 && PyErr_Occurred()) return -1;
    *out = (void*)(uintptr_t)v;
    return 0;
}

/* bind(addr_of_gw_step, addr_of_gw_pendulum_step, addr_of_gw_step_fb) */
static PyObject* py_bind(PyObject* self, PyObject* const* args, Py_ssize_t n)
{
    void *a = NULL, *b = NULL, *c = NULL;
    if (n != 3) { PyErr_SetString(PyExc_TypeError, "bind(gw_step, gw_pendulum_step, gw_step_fb)"); return NULL; }
    if (as_ptr(args[0], &a) || as_ptr(args[1], &b) || as_ptr(args[2], &c)) return NULL;
    g_step = (gw_step_fn)a;
    g_pend = (gw_pend_fn)b;
    g_step_fb = (gw_step_fb_fn)c;
    Py_RETURN_NONE;
}

/* step_fb(env, device, duration, obs, reward, done, feedback_byte, stream) -> rc */
static PyObject* py_step_fb(PyObject* self, PyObject* const* args, Py_ssize_t n)
{
    void* p[8];
    if (n != 8 || !g_step_fb) { PyErr_SetString(PyExc_TypeError, "step_fb(env, device, duration, obs, reward, done, feedback_byte, stream) after bind()"); return NULL; }
    for (int i = 0; i < 8; ++i)
        if (as_ptr(args[i], &p[i])) return NULL;
    const int rc = g_step_fb(p[0], (const int32_t*)p[1], (const int32_t*)p[2], (int32_t*)p[3], (float*)p[4], (uint8_t*)p[5], (uint8_t*)p[6], p[7]);
    return PyLong_FromLong(rc);
}

/* step(env, device, duration, obs, reward, done, stream) -> rc; all arguments are addresses as Python ints */
static PyObject* py_step(PyObject* self, PyObject* const* args, Py_ssize_t n)
{
    void* p[7];
    if (n != 7 || !g_step) { PyErr_SetString(PyExc_TypeError, "step(env, device, duration, obs, reward, done, stream) after bind()"); return NULL; }
    for (int i = 0; i < 7; ++i)
        if (as_ptr(args[i], &p[i])) return NULL;
    const int rc = g_step(p[0], (const int32_t*)p[1], (const int32_t*)p[2], (int32_t*)p[3], (float*)p[4], (uint8_t*)p[5], p[6]);
    return PyLong_FromLong(rc);
}

/* pendulum_step(env, plant, device, duration, obs, reward, angle_deg, stream) -> rc */
static PyObject* py_pend(PyObject* self, PyObject* const* args, Py_ssize_t n)
{
    void* p[8];
    if (n != 8 || !g_pend) { PyErr_SetString(PyExc_TypeError, "pendulum_step(env, plant, device, duration, obs, reward, angle, stream) after bind()"); return NULL; }
    for (int i = 0; i < 8; ++i)
        if (as_ptr(args[i], &p[i])) return NULL;
    const int rc = g_pend(p[0], p[1], (const int32_t*)p[2], (const int32_t*)p[3], (int32_t*)p[4], (float*)p[5], (double*)p[6], p[7]);
    return PyLong_FromLong(rc);
}

static PyMethodDef methods[] = {
    {"bind", (PyCFunction)(void (*)(void))py_bind, METH_FASTCALL, "bind(gw_step address, gw_pendulum_step address, gw_step_fb address)"},
    {"step", (PyCFunction)(void (*)(void))py_step, METH_FASTCALL, "gw_step with addresses as ints"},
    {"step_fb", (PyCFunction)(void (*)(void))py_step_fb, METH_FASTCALL, "gw_step_fb with addresses as ints"},
    {"pendulum_step", (PyCFunction)(void (*)(void))py_pend, METH_FASTCALL, "gw_pendulum_step with addresses as ints"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_gw_fast", "fast-call shim for the per-step C-ABI entry points", -1, methods};

PyMODINIT_FUNC PyInit__gw_fast(void) { return PyModule_Create(&moddef); }
