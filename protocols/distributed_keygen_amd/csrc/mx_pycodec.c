/* Python int  <->  little-endian uint32 limb rows, in bulk (CPython extension, host side only).
 *
 * The C ABI's number format is `int.to_bytes(4*limbs, "little")` per element (include/mxpaillier.h).
 * Converting 10 000 ciphertexts of 4100 bits with a Python-level loop costs 12 ms in and 12 ms out —
 * a fifth of the GPU time of the whole batch; this module does the same conversion in one C loop per
 * batch (the reference's values are Python ints: PaillierCiphertext.get_value(), PSK:69; the results
 * go back as Python ints, PSK:92 / PSK:125).  No arithmetic happens here.
 *
 *   pack_into(values, limbs, buffer, row_offset) -> None   buffer: writable, C-contiguous, rows of 4*limbs bytes
 *   unpack(buffer, limbs) -> list[int]
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <string.h>

static PyObject* pack_into(PyObject* self, PyObject* args) {
  PyObject* seq;
  Py_ssize_t limbs, row_offset;
  Py_buffer out;
  if (!PyArg_ParseTuple(args, "Onw*n", &seq, &limbs, &out, &row_offset)) return NULL;
  PyObject* fast = PySequence_Fast(seq, "values must be a sequence of ints");
  if (!fast) { PyBuffer_Release(&out); return NULL; }
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
  const Py_ssize_t nbytes = 4 * limbs;
  if (limbs <= 0 || row_offset < 0 || (row_offset + n) * nbytes > out.len) {
    PyErr_SetString(PyExc_ValueError, "buffer too small for the rows");
    goto fail;
  }
  unsigned char* dst = (unsigned char*)out.buf + row_offset * nbytes;
  PyObject** items = PySequence_Fast_ITEMS(fast);
  for (Py_ssize_t i = 0; i < n; ++i, dst += nbytes) {
    PyObject* v = items[i];
    if (!PyLong_Check(v)) {
      PyErr_SetString(PyExc_TypeError, "values must be ints");
      goto fail;
    }
    /* unsigned, little endian; fails (OverflowError) for negative values and values that do not fit */
    if (_PyLong_AsByteArray((PyLongObject*)v, dst, (size_t)nbytes, 1, 0) < 0) {
      PyErr_Clear();
      PyErr_Format(PyExc_ValueError, "value does not fit in %zd uint32 limbs (or is negative)", limbs);
      goto fail;
    }
  }
  Py_DECREF(fast);
  PyBuffer_Release(&out);
  Py_RETURN_NONE;
fail:
  Py_DECREF(fast);
  PyBuffer_Release(&out);
  return NULL;
}

static PyObject* unpack(PyObject* self, PyObject* args) {
  Py_buffer in;
  Py_ssize_t limbs;
  if (!PyArg_ParseTuple(args, "y*n", &in, &limbs)) return NULL;
  const Py_ssize_t nbytes = 4 * limbs;
  if (limbs <= 0 || in.len % nbytes != 0) {
    PyBuffer_Release(&in);
    PyErr_SetString(PyExc_ValueError, "buffer is not a whole number of rows");
    return NULL;
  }
  const Py_ssize_t n = in.len / nbytes;
  PyObject* list = PyList_New(n);
  if (!list) { PyBuffer_Release(&in); return NULL; }
  const unsigned char* src = (const unsigned char*)in.buf;
  for (Py_ssize_t i = 0; i < n; ++i, src += nbytes) {
    PyObject* v = _PyLong_FromByteArray(src, (size_t)nbytes, 1, 0);
    if (!v) { Py_DECREF(list); PyBuffer_Release(&in); return NULL; }
    PyList_SET_ITEM(list, i, v);
  }
  PyBuffer_Release(&in);
  return list;
}

static PyMethodDef methods[] = {
    {"pack_into", pack_into, METH_VARARGS, "pack_into(values, limbs, buffer, row_offset): ints -> uint32 rows"},
    {"unpack", unpack, METH_VARARGS, "unpack(buffer, limbs) -> list of ints"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_mxcodec", "bulk Python int <-> limb rows", -1, methods};

PyMODINIT_FUNC PyInit__mxcodec(void) { return PyModule_Create(&moduledef); }
