/* The module the reference imports as `exllamav3_ext` (exllamav3/ext.py:20-30 only accepts a spec whose origin ends in an extension suffix, so a
 * .py shim is not enough): a CPython extension whose init imports exllamav3_amd.ext -- the Python mirror of the reference op surface over the
 * C-ABI library libexl3_hip.so -- and re-exports its namespace.  Names outside the hot path resolve through the mirror's module __getattr__
 * (an AttributeError naming the op).  No torch, no HIP in here.  Built by __graft_entry__.build() into exllamav3_amd/stub/ (put that directory
 * on sys.path / PYTHONPATH, or copy the file next to the reference package). */
#define PY_SSIZE_T_CLEAN
#include <Python.h>

static PyObject* stub_getattr(PyObject* self, PyObject* name)
{
    (void) self;
    PyObject* impl = PyImport_ImportModule("exllamav3_amd.ext");
    if (!impl) return NULL;
    PyObject* r = PyObject_GetAttr(impl, name);
    Py_DECREF(impl);
    return r;
}

static PyMethodDef stub_methods[] = {
    { "__getattr__", stub_getattr, METH_O, "late lookups are forwarded to exllamav3_amd.ext" },
    { NULL, NULL, 0, NULL }
};

static struct PyModuleDef stub_def = {
    PyModuleDef_HEAD_INIT, "exllamav3_ext",
    "MI355X-native implementation of the exllamav3_ext op surface for the EXL3 quantized-linear hot path (exllamav3_amd.ext over libexl3_hip.so)",
    -1, stub_methods, NULL, NULL, NULL, NULL
};

PyMODINIT_FUNC PyInit_exllamav3_ext(void)
{
    PyObject* impl = PyImport_ImportModule("exllamav3_amd.ext");
    if (!impl) return NULL;
    PyObject* m = PyModule_Create(&stub_def);
    if (!m) { Py_DECREF(impl); return NULL; }
    PyObject* src = PyModule_GetDict(impl);      /* borrowed */
    PyObject* dst = PyModule_GetDict(m);         /* borrowed */
    PyObject *key, *value;
    Py_ssize_t pos = 0;
    while (PyDict_Next(src, &pos, &key, &value))
    {
        /* public names only: the stub keeps its own __name__ / __spec__ / __loader__ / __file__ */
        if (PyUnicode_Check(key) && PyUnicode_GET_LENGTH(key) >= 2 && PyUnicode_READ_CHAR(key, 0) == '_' && PyUnicode_READ_CHAR(key, 1) == '_') continue;
        if (PyDict_SetItem(dst, key, value) < 0) { Py_DECREF(impl); Py_DECREF(m); return NULL; }
    }
    if (PyModule_AddObject(m, "__implementation__", impl) < 0) { Py_DECREF(impl); Py_DECREF(m); return NULL; }   /* steals impl */
    return m;
}
