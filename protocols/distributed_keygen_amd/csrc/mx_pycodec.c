/* Python int  <->  little-endian uint32 limb rows, in bulk (CPython extension, host side only).
 *
 * The C ABI's number format is `int.to_bytes(4*limbs, "little")` per element (include/mxpaillier.h).
 * The reference's values are Python ints (PaillierCiphertext.get_value(), PSK:69; the shares, generators and v values
 * of a key-generation round, DK:1284-1360) and its results go back as Python ints (PSK:92 / PSK:125), so every
 * int-level call of the engine converts whole batches: 10 000 ciphertexts of 4100 bits, or the 230 000 generators
 * and 290 000 v values of a 65 536-candidate round — two thirds of such a round's time in round 3, when this module
 * called _PyLong_AsByteArray per element.  No arithmetic happens here.
 *
 *   pack_into(values, limbs, buffer, row_offset) -> None   buffer: writable, C-contiguous, rows of 4*limbs bytes
 *   unpack(buffer, limbs) -> list[int]
 *   pack_nested_into(lists, inner, limbs, buffer, row_offset) -> None    lists[g] contributes `inner` rows: its first
 *                                                           min(len, inner) ints, then zero rows — the generator lists and
 *                                                           v lists of a key-generation round (one list per candidate,
 *                                                           DK:1313-1360) packed as the reference holds them, without a
 *                                                           flattened copy of 230 000 references first
 *   unpack_groups(buffer, limbs, counts, stride) -> list[list[int]]     group g = rows [g * stride, g * stride + counts[g]):
 *                                                           the v lists of DK:1103-1108 built in one pass (no flat list
 *                                                           that is sliced — and every int touched — a second time)
 *   rows_ge(rows, limbs, moduli_rows, group) -> list[int]  indices of rows that are >= their group's modulus
 *   set_threads(n) -> previous setting                     0 = automatic (usable cores, at most 16)
 *
 * Both directions read / write the digits of the int objects directly (CPython's 30-bit digits) and do the bit
 * shuffling on several threads that never touch the Python API.  Packing keeps the interpreter lock in the calling
 * thread (nobody can change the list or free an element meanwhile: no per-element reference counting, see run_pack);
 * unpacking allocates the new ints under the lock (sizes computed first, in parallel, without it) and fills them in
 * parallel without it, before anybody else can see them.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <string.h>
#include <unistd.h>

#if PYLONG_BITS_IN_DIGIT == 30 && PY_VERSION_HEX < 0x030C0000
#define MX_DIRECT_DIGITS 1
#else
#define MX_DIRECT_DIGITS 0
#endif

static int g_threads = 0; /* 0 = automatic */

static int usable_threads(Py_ssize_t n, Py_ssize_t per_thread_min) {
  int t = g_threads;
  if (t <= 0) {
    cpu_set_t set;
    t = (sched_getaffinity(0, sizeof(set), &set) == 0) ? CPU_COUNT(&set) : (int)sysconf(_SC_NPROCESSORS_ONLN);
    if (t > 16) t = 16; /* the GPU boxes report 256 CPUs and grant 16 */
  }
  if (t < 1) t = 1;
  if ((Py_ssize_t)t > n / per_thread_min) t = (int)(n / per_thread_min);
  return t < 1 ? 1 : t;
}

#if MX_DIRECT_DIGITS
/* ---------------------------------------------------------------------------------------------- pack */
typedef struct { PyObject** items; uint32_t* dst; Py_ssize_t limbs, lo, hi; Py_ssize_t bad; int bad_kind; } PackJob;

static void digits_to_words(const digit* d, Py_ssize_t nd, uint32_t* out, Py_ssize_t limbs) {
  uint64_t acc = 0;
  int bits = 0;
  Py_ssize_t w = 0;
  for (Py_ssize_t i = 0; i < nd; ++i) {
    acc |= (uint64_t)d[i] << bits;
    bits += 30;
    if (bits >= 32) {
      if (w < limbs) out[w] = (uint32_t)acc;
      ++w;
      acc >>= 32;
      bits -= 32;
    }
  }
  if (w < limbs) out[w++] = (uint32_t)acc;
  for (; w < limbs; ++w) out[w] = 0;
}

/* Everything about an element happens here, on the worker's core (the object header and its digits share cache
 * lines: touching them first on the calling thread would only move the misses there): type, sign, size, conversion.
 * Only immutable fields of live objects are read; a failure is recorded (first one of the slice) and raised by the
 * caller under the lock. */
static void* pack_worker(void* arg) {
  PackJob* j = (PackJob*)arg;
  const Py_ssize_t room = 32 * j->limbs;
  for (Py_ssize_t i = j->lo; i < j->hi; ++i) {
    PyObject* v = j->items[i];
    if (!v) { memset(j->dst + i * j->limbs, 0, (size_t)j->limbs * 4); continue; }      /* padding row (pack_nested_into) */
    if (!PyLong_Check(v)) { j->bad = i; j->bad_kind = 1; return NULL; }
    const Py_ssize_t nd = Py_SIZE(v);
    const digit* d = ((PyLongObject*)v)->ob_digit;
    int fits = nd >= 0;
    if (fits && nd > 0) {
      Py_ssize_t bits = (nd - 1) * 30 + (32 - __builtin_clz((unsigned)d[nd - 1]));
      fits = bits <= room;
    }
    if (!fits) { j->bad = i; j->bad_kind = 2; return NULL; }
    digits_to_words(d, nd, j->dst + i * j->limbs, j->limbs);
  }
  return NULL;
}

/* items[0..n) (NULL = a zero row) -> rows at dst, on several threads.  The CALLER KEEPS THE INTERPRETER LOCK for the whole
 * call: the worker threads call nothing of the Python API — they read the size and the digits of immutable int objects —
 * and while this thread holds the lock no other Python thread can run, so nobody can resize or overwrite the list the
 * pointers came from or drop the last reference to an element.  (Rounds 4-5 released the lock and therefore took a
 * reference to every element first and gave it back afterwards: two single-threaded passes that write to the header of
 * every one of the 330 000 ints of a key-generation round — more than the conversion itself cost.)  Other Python threads
 * are held up for the milliseconds the conversion takes, like behind any C call that does not release the lock.
 * Returns 1, or 0 with a Python exception set. */
static int run_pack(PyObject** items, Py_ssize_t n, uint32_t* dst, Py_ssize_t limbs) {
  const int nt = usable_threads(n, 1024);
  PackJob jobs[16];
  pthread_t tid[16];
  int started[16] = {0};
  for (int t = 0; t < nt; ++t) {
    jobs[t].items = items; jobs[t].dst = dst; jobs[t].limbs = limbs;
    jobs[t].lo = n * t / nt; jobs[t].hi = n * (t + 1) / nt; jobs[t].bad = -1; jobs[t].bad_kind = 0;
    if (t > 0) started[t] = pthread_create(&tid[t], NULL, pack_worker, &jobs[t]) == 0;
  }
  pack_worker(&jobs[0]);
  for (int t = 1; t < nt; ++t) {
    if (started[t]) pthread_join(tid[t], NULL);
    else pack_worker(&jobs[t]); /* thread creation failed: this thread does the slice */
  }
  for (int t = 0; t < nt; ++t) {
    if (jobs[t].bad_kind == 1) { PyErr_SetString(PyExc_TypeError, "values must be ints"); return 0; }
    if (jobs[t].bad_kind == 2) {
      PyErr_Format(PyExc_ValueError, "value does not fit in %zd uint32 limbs (or is negative)", limbs);
      return 0;
    }
  }
  return 1;
}

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
  if (((uintptr_t)out.buf & 3u) != 0) {
    PyErr_SetString(PyExc_ValueError, "rows must be 4-byte aligned");
    goto fail;
  }
  {
    uint32_t* dst = (uint32_t*)((unsigned char*)out.buf + row_offset * nbytes);
    /* the element pointers of the list itself: it cannot change while this thread holds the lock (run_pack) */
    const int ok = run_pack(PySequence_Fast_ITEMS(fast), n, dst, limbs);
    if (!ok) goto fail;
  }
  Py_DECREF(fast);
  PyBuffer_Release(&out);
  Py_RETURN_NONE;
fail:
  Py_DECREF(fast);
  PyBuffer_Release(&out);
  return NULL;
}

/* pack_nested_into(lists, inner, limbs, buffer, row_offset): lists[g] -> rows [row_offset + g * inner, ... + inner) */
static PyObject* pack_nested_into(PyObject* self, PyObject* args) {
  PyObject* seq;
  Py_ssize_t inner, limbs, row_offset;
  Py_buffer out;
  if (!PyArg_ParseTuple(args, "Onnw*n", &seq, &inner, &limbs, &out, &row_offset)) return NULL;
  PyObject* fast = PySequence_Fast(seq, "lists must be a sequence of sequences of ints");
  if (!fast) { PyBuffer_Release(&out); return NULL; }
  const Py_ssize_t groups = PySequence_Fast_GET_SIZE(fast);
  const Py_ssize_t nbytes = 4 * limbs;
  PyObject** items = NULL;
  PyObject* keep = NULL;
  Py_ssize_t n = 0;
  int ok = 0;
  if (limbs <= 0 || inner <= 0 || row_offset < 0 || groups > (PY_SSIZE_T_MAX / 8) / inner || (row_offset + groups * inner) * nbytes > out.len) {
    PyErr_SetString(PyExc_ValueError, "buffer too small for the rows");
    goto done;
  }
  if (((uintptr_t)out.buf & 3u) != 0) {
    PyErr_SetString(PyExc_ValueError, "rows must be 4-byte aligned");
    goto done;
  }
  n = groups * inner;
  items = (PyObject**)PyMem_Calloc((size_t)(n ? n : 1), sizeof(PyObject*));
  if (!items) { PyErr_NoMemory(); goto done; }
  for (Py_ssize_t g = 0; g < groups; ++g) {
    PyObject* in = PySequence_Fast(PySequence_Fast_GET_ITEM(fast, g), "lists must be a sequence of sequences of ints");
    if (!in) goto done;
    Py_ssize_t k = PySequence_Fast_GET_SIZE(in);
    if (k > inner) k = inner;
    PyObject** src = PySequence_Fast_ITEMS(in);
    /* borrowed pointers: `in` is the inner list / tuple itself, which the outer sequence keeps alive while this thread
     * holds the lock — or, for another kind of sequence (an iterator's values exist nowhere else), a NEW list, which is
     * kept until the rows are written */
    for (Py_ssize_t i = 0; i < k; ++i) items[g * inner + i] = src[i];
    if (in != PySequence_Fast_GET_ITEM(fast, g)) {
      if (!keep) keep = PyList_New(0);
      if (!keep || PyList_Append(keep, in) < 0) { Py_DECREF(in); goto done; }
    }
    Py_DECREF(in);
  }
  ok = run_pack(items, n, (uint32_t*)((unsigned char*)out.buf + row_offset * nbytes), limbs);
done:
  PyMem_Free(items);
  Py_XDECREF(keep);
  Py_DECREF(fast);
  PyBuffer_Release(&out);
  if (!ok) return NULL;
  Py_RETURN_NONE;
}

/* -------------------------------------------------------------------------------------------- unpack */
typedef struct { const uint32_t* rows; Py_ssize_t limbs, lo, hi; Py_ssize_t* nd; PyObject** objs; int fill; const Py_ssize_t* src; } UnpackJob;

static void* unpack_worker(void* arg) {
  UnpackJob* j = (UnpackJob*)arg;
  for (Py_ssize_t i = j->lo; i < j->hi; ++i) {
    const uint32_t* w = j->rows + (j->src ? j->src[i] : i) * j->limbs;      /* src: the row of output element i (unpack_groups) */
    if (!j->fill) { /* pass 1: digits needed */
      Py_ssize_t top = j->limbs;
      while (top > 0 && w[top - 1] == 0) --top;
      Py_ssize_t bits = 0;
      if (top > 0) {
        bits = 32 * (top - 1);
        uint32_t x = w[top - 1];
        while (x) { ++bits; x >>= 1; }
      }
      j->nd[i] = (bits + 29) / 30;
    } else if (j->nd[i] > 0) { /* pass 2: words -> 30-bit digits of the freshly allocated int */
      digit* d = ((PyLongObject*)j->objs[i])->ob_digit;
      const Py_ssize_t nd = j->nd[i];
      uint64_t acc = 0;
      int bits = 0;
      Py_ssize_t k = 0, wi = 0;
      while (k < nd) {
        if (bits < 30) {                       /* at most 29 + 32 bits in the accumulator */
          if (wi < j->limbs) acc |= (uint64_t)w[wi++] << bits;
          bits += 32;
        }
        d[k++] = (digit)(acc & 0x3FFFFFFFu);
        acc >>= 30;
        bits -= 30;
      }
    }
  }
  return NULL;
}

static void run_unpack(UnpackJob* proto, Py_ssize_t n, int nt) {
  UnpackJob jobs[16];
  pthread_t tid[16];
  int started[16] = {0};
  for (int t = 0; t < nt; ++t) {
    jobs[t] = *proto;
    jobs[t].lo = n * t / nt; jobs[t].hi = n * (t + 1) / nt;
    if (t > 0) started[t] = pthread_create(&tid[t], NULL, unpack_worker, &jobs[t]) == 0;
  }
  unpack_worker(&jobs[0]);
  for (int t = 1; t < nt; ++t) {
    if (started[t]) pthread_join(tid[t], NULL);
    else unpack_worker(&jobs[t]);
  }
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
  Py_ssize_t* nd = (Py_ssize_t*)PyMem_Malloc((size_t)(n ? n : 1) * sizeof(Py_ssize_t));
  PyObject** objs = (PyObject**)PyMem_Calloc((size_t)(n ? n : 1), sizeof(PyObject*));
  if (!list || !nd || !objs || ((uintptr_t)in.buf & 3)) {
    if (list && nd && objs) PyErr_SetString(PyExc_ValueError, "rows must be 4-byte aligned");
    else PyErr_NoMemory();
    goto fail;
  }
  const int nt = usable_threads(n, 2048);
  UnpackJob job = {(const uint32_t*)in.buf, limbs, 0, 0, nd, objs, 0, NULL};
  Py_BEGIN_ALLOW_THREADS
  run_unpack(&job, n, nt);
  Py_END_ALLOW_THREADS
  for (Py_ssize_t i = 0; i < n; ++i) {
    objs[i] = nd[i] == 0 ? PyLong_FromLong(0) : (PyObject*)_PyLong_New(nd[i]);
    if (!objs[i]) goto fail;
  }
  job.fill = 1;
  Py_BEGIN_ALLOW_THREADS
  run_unpack(&job, n, nt);
  Py_END_ALLOW_THREADS
  for (Py_ssize_t i = 0; i < n; ++i) PyList_SET_ITEM(list, i, objs[i]);
  PyMem_Free(nd);
  PyMem_Free(objs);
  PyBuffer_Release(&in);
  return list;
fail:
  if (objs) for (Py_ssize_t i = 0; i < n; ++i) Py_XDECREF(objs[i]);
  PyMem_Free(nd);
  PyMem_Free(objs);
  Py_XDECREF(list);
  PyBuffer_Release(&in);
  return NULL;
}

/* unpack_groups(buffer, limbs, counts, stride) -> [[int, ...], ...]: group g = rows [g * stride, g * stride + counts[g]) */
static PyObject* unpack_groups(PyObject* self, PyObject* args) {
  Py_buffer in;
  Py_ssize_t limbs, stride;
  PyObject* counts_obj;
  if (!PyArg_ParseTuple(args, "y*nOn", &in, &limbs, &counts_obj, &stride)) return NULL;
  PyObject* counts = PySequence_Fast(counts_obj, "counts must be a sequence of ints");
  if (!counts) { PyBuffer_Release(&in); return NULL; }
  const Py_ssize_t nbytes = 4 * limbs;
  const Py_ssize_t groups = PySequence_Fast_GET_SIZE(counts);
  PyObject* outer = NULL;
  Py_ssize_t* nd = NULL;
  Py_ssize_t* src = NULL;
  Py_ssize_t* cnt = NULL;
  PyObject** objs = NULL;
  Py_ssize_t n = 0;
  int ok = 0;
  if (limbs <= 0 || stride <= 0 || in.len % nbytes != 0 || ((uintptr_t)in.buf & 3)) {
    PyErr_SetString(PyExc_ValueError, "buffer must be whole, 4-byte aligned rows");
    goto done;
  }
  cnt = (Py_ssize_t*)PyMem_Malloc((size_t)(groups ? groups : 1) * sizeof(Py_ssize_t));
  if (!cnt) { PyErr_NoMemory(); goto done; }
  for (Py_ssize_t g = 0; g < groups; ++g) {
    const Py_ssize_t c = PyNumber_AsSsize_t(PySequence_Fast_GET_ITEM(counts, g), PyExc_OverflowError);
    if (c == -1 && PyErr_Occurred()) goto done;
    if (c < 0 || c > stride) { PyErr_SetString(PyExc_ValueError, "a count must lie in 0 .. stride"); goto done; }
    cnt[g] = c;
    n += c;
  }
  if (groups * stride * nbytes > in.len) { PyErr_SetString(PyExc_ValueError, "buffer holds fewer than groups x stride rows"); goto done; }
  nd = (Py_ssize_t*)PyMem_Malloc((size_t)(n ? n : 1) * sizeof(Py_ssize_t));
  src = (Py_ssize_t*)PyMem_Malloc((size_t)(n ? n : 1) * sizeof(Py_ssize_t));
  objs = (PyObject**)PyMem_Calloc((size_t)(n ? n : 1), sizeof(PyObject*));
  outer = PyList_New(groups);
  if (!nd || !src || !objs || !outer) { PyErr_NoMemory(); goto done; }
  {
    Py_ssize_t i = 0;
    for (Py_ssize_t g = 0; g < groups; ++g)
      for (Py_ssize_t k = 0; k < cnt[g]; ++k) src[i++] = g * stride + k;
  }
  {
    const int nt = usable_threads(n, 2048);
    UnpackJob job = {(const uint32_t*)in.buf, limbs, 0, 0, nd, objs, 0, src};
    Py_BEGIN_ALLOW_THREADS
    run_unpack(&job, n, nt);
    Py_END_ALLOW_THREADS
    for (Py_ssize_t i = 0; i < n; ++i) {
      objs[i] = nd[i] == 0 ? PyLong_FromLong(0) : (PyObject*)_PyLong_New(nd[i]);
      if (!objs[i]) goto done;
    }
    job.fill = 1;
    Py_BEGIN_ALLOW_THREADS
    run_unpack(&job, n, nt);
    Py_END_ALLOW_THREADS
  }
  {
    Py_ssize_t i = 0;
    for (Py_ssize_t g = 0; g < groups; ++g) {
      PyObject* inner = PyList_New(cnt[g]);
      if (!inner) goto done;
      for (Py_ssize_t k = 0; k < cnt[g]; ++k, ++i) { PyList_SET_ITEM(inner, k, objs[i]); objs[i] = NULL; }      /* the reference moves */
      PyList_SET_ITEM(outer, g, inner);
    }
  }
  ok = 1;
done:
  if (objs) for (Py_ssize_t i = 0; i < n; ++i) Py_XDECREF(objs[i]);
  PyMem_Free(nd);
  PyMem_Free(src);
  PyMem_Free(cnt);
  PyMem_Free(objs);
  Py_DECREF(counts);
  PyBuffer_Release(&in);
  if (!ok) { Py_XDECREF(outer); return NULL; }
  return outer;
}

#else /* ---- interpreters with another digit width or int layout: the per-element library routines */

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
  if (((uintptr_t)out.buf & 3u) != 0) {
    PyErr_SetString(PyExc_ValueError, "rows must be 4-byte aligned");
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
#endif

typedef struct { const uint32_t* rows; const uint32_t* mods; Py_ssize_t limbs, group, lo, hi; Py_ssize_t* found; Py_ssize_t nfound, cap; int oom; } GeJob;

static void* ge_worker(void* arg) {
  GeJob* j = (GeJob*)arg;
  for (Py_ssize_t e = j->lo; e < j->hi; ++e) {
    const uint32_t* a = j->rows + e * j->limbs;
    const uint32_t* b = j->mods + (e / j->group) * j->limbs;
    int ge = 1; /* equal counts as >= */
    for (Py_ssize_t w = j->limbs - 1; w >= 0; --w) {
      if (a[w] != b[w]) { ge = a[w] > b[w]; break; }
    }
    if (ge) {
      if (j->nfound == j->cap) {
        const Py_ssize_t cap = j->cap ? 2 * j->cap : 64;
        Py_ssize_t* grown = (Py_ssize_t*)realloc(j->found, (size_t)cap * sizeof(Py_ssize_t));
        if (!grown) { j->oom = 1; return NULL; }
        j->found = grown; j->cap = cap;
      }
      j->found[j->nfound++] = e;
    }
  }
  return NULL;
}

/* rows_ge(rows, limbs, moduli_rows, group) -> list of the indices e with rows[e] >= moduli_rows[e / group]
 * (both buffers little-endian uint32 rows of `limbs` words).  Received values are canonical residues already, so the
 * list is normally empty: this replaces a per-group Python / numpy pass over every row in front of each launch. */
static PyObject* rows_ge(PyObject* self, PyObject* args) {
  Py_buffer rows, mods;
  Py_ssize_t limbs, group;
  if (!PyArg_ParseTuple(args, "y*ny*n", &rows, &limbs, &mods, &group)) return NULL;
  PyObject* out = NULL;
  const Py_ssize_t nbytes = 4 * limbs;
  if (limbs <= 0 || group <= 0 || rows.len % nbytes != 0 || mods.len % nbytes != 0 || (((uintptr_t)rows.buf | (uintptr_t)mods.buf) & 3)) {
    PyErr_SetString(PyExc_ValueError, "rows_ge: buffers must be whole, 4-byte aligned rows");
    goto done;
  }
  {
    const Py_ssize_t n = rows.len / nbytes, ngroups = mods.len / nbytes;
    if (n > ngroups * group) {
      PyErr_SetString(PyExc_ValueError, "rows_ge: more rows than moduli x group");
      goto done;
    }
    out = PyList_New(0);
    if (!out) goto done;
    /* the comparison on several threads (no Python API inside); a thread notes the rows it finds — normally none — in its
     * own small array, and the list is built here, in ascending order */
    GeJob jobs[16];
    pthread_t tid[16];
    int started[16] = {0};
    const int nt = usable_threads(n, 8192);
    Py_BEGIN_ALLOW_THREADS
    for (int t = 0; t < nt; ++t) {
      jobs[t].rows = (const uint32_t*)rows.buf; jobs[t].mods = (const uint32_t*)mods.buf; jobs[t].limbs = limbs; jobs[t].group = group;
      jobs[t].lo = n * t / nt; jobs[t].hi = n * (t + 1) / nt; jobs[t].found = NULL; jobs[t].nfound = 0; jobs[t].cap = 0; jobs[t].oom = 0;
      if (t > 0) started[t] = pthread_create(&tid[t], NULL, ge_worker, &jobs[t]) == 0;
    }
    ge_worker(&jobs[0]);
    for (int t = 1; t < nt; ++t) {
      if (started[t]) pthread_join(tid[t], NULL);
      else ge_worker(&jobs[t]);
    }
    Py_END_ALLOW_THREADS
    int failed = 0;
    for (int t = 0; t < nt; ++t) {
      if (jobs[t].oom) failed = 1;
      for (Py_ssize_t k = 0; k < jobs[t].nfound && !failed; ++k) {
        PyObject* idx = PyLong_FromSsize_t(jobs[t].found[k]);
        if (!idx || PyList_Append(out, idx) < 0) failed = 1;
        Py_XDECREF(idx);
      }
      free(jobs[t].found);
    }
    if (failed) { if (!PyErr_Occurred()) PyErr_NoMemory(); Py_CLEAR(out); }
  }
done:
  PyBuffer_Release(&rows);
  PyBuffer_Release(&mods);
  return out;
}

static PyObject* set_threads(PyObject* self, PyObject* args) {
  int n;
  if (!PyArg_ParseTuple(args, "i", &n)) return NULL;
  if (n < 0 || n > 16) {
    PyErr_SetString(PyExc_ValueError, "threads must be 0 (automatic) .. 16");
    return NULL;
  }
  const int prev = g_threads;
  g_threads = n;
  return PyLong_FromLong(prev);
}

static PyMethodDef methods[] = {
    {"pack_into", pack_into, METH_VARARGS, "pack_into(values, limbs, buffer, row_offset): ints -> uint32 rows"},
    {"unpack", unpack, METH_VARARGS, "unpack(buffer, limbs) -> list of ints"},
#if MX_DIRECT_DIGITS
    {"pack_nested_into", pack_nested_into, METH_VARARGS, "pack_nested_into(lists, inner, limbs, buffer, row_offset): inner rows per list, zero padded"},
    {"unpack_groups", unpack_groups, METH_VARARGS, "unpack_groups(buffer, limbs, counts, stride) -> list of lists of ints"},
#endif
    {"rows_ge", rows_ge, METH_VARARGS, "rows_ge(rows, limbs, moduli_rows, group) -> indices of rows >= their modulus"},
    {"set_threads", set_threads, METH_VARARGS, "set_threads(n) -> previous; 0 = automatic (usable cores, at most 16)"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_mxcodec", "bulk Python int <-> limb rows", -1, methods};

PyMODINIT_FUNC PyInit__mxcodec(void) {
  PyObject* m = PyModule_Create(&moduledef);
  if (m) PyModule_AddIntConstant(m, "DIRECT_DIGITS", MX_DIRECT_DIGITS);
  return m;
}
