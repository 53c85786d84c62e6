"""Loads the NIF shims of c_src/ without a BEAM: builds tests/host/fake_erl_nif.c (the enif_*
functions declared in c_src/erl_nif_decl.h over a flat term table) and the two shims against
libexmc_hip.so, and converts between Python values and the fake runtime's terms. Test
infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "exmc_amd", "lib")
T_ATOM, T_DOUBLE, T_INT, T_BIN, T_LIST, T_TUPLE, T_MAP, T_RES, T_STR = range(1, 10)


class FuncEntry(C.Structure):
    _fields_ = [("name", C.c_char_p), ("arity", C.c_uint), ("fptr", C.c_void_p), ("flags", C.c_uint)]


class Entry(C.Structure):
    _fields_ = [("major", C.c_int), ("minor", C.c_int), ("name", C.c_char_p),
                ("num_of_funcs", C.c_int), ("funcs", C.POINTER(FuncEntry)), ("load", C.c_void_p),
                ("reload", C.c_void_p), ("upgrade", C.c_void_p), ("unload", C.c_void_p),
                ("vm_variant", C.c_char_p), ("options", C.c_uint),
                ("sizeof_ErlNifResourceTypeInit", C.c_size_t), ("min_erts", C.c_char_p)]


class Atom(str):
    pass


class Raised(Exception):
    def __init__(self, reason):
        super().__init__(repr(reason))
        self.reason = reason


class BadArg(Exception):
    pass


_cache = {}


def build(outdir):
    # ONE term table per process: the fake runtime is loaded RTLD_GLOBAL (the shims resolve their
    # enif_* calls against it), so a second copy would split the state between two libraries
    if _cache:
        return next(iter(_cache.values()))
    fake = os.path.join(outdir, "libfake_erl_nif.so")
    # EXMC_SANITIZE=address,undefined (tools/sanitize_cpu.sh, with libasan preloaded into the interpreter):
    # the shims and the term table are built with the sanitizers
    san = os.environ.get("EXMC_SANITIZE")
    sflags = ["-fsanitize=" + san, "-fno-omit-frame-pointer", "-g"] if san else []
    subprocess.check_call(["gcc", "-std=gnu11", "-O1", "-Wall", "-Wextra", "-fPIC", "-shared", "-pthread"] + sflags +
                          ["-o", fake, os.path.join(ROOT, "tests", "host", "fake_erl_nif.c")])
    shims = {}
    for mod, src in (("NativeTree", "exmc_native_tree_nif.c"), ("HipNative", "exmc_hip_nif.c")):
        so = os.path.join(outdir, "lib%s_nif.so" % mod)
        subprocess.check_call(["gcc", "-std=c11", "-O2", "-Wall", "-Wextra", "-Werror", "-fPIC", "-shared"] + sflags +
                              ["-o", so, os.path.join(ROOT, "c_src", src), "-L" + LIBDIR, "-lexmc_hip",
                               "-Wl,-rpath," + LIBDIR, "-Wl,-z,lazy", "-ldl"])
        shims[mod] = so
    F = C.CDLL(fake, mode=os.RTLD_GLOBAL | os.RTLD_NOW)
    tt = C.c_ulong
    F.fk_atom.argtypes = [C.c_char_p]; F.fk_atom.restype = tt
    F.fk_double.argtypes = [C.c_double]; F.fk_double.restype = tt
    F.fk_int.argtypes = [C.c_longlong]; F.fk_int.restype = tt
    F.fk_binary.argtypes = [C.c_void_p, C.c_size_t]; F.fk_binary.restype = tt
    F.fk_list.argtypes = [C.POINTER(tt), C.c_uint]; F.fk_list.restype = tt
    F.fk_type.argtypes = [tt]
    F.fk_get_double.argtypes = [tt]; F.fk_get_double.restype = C.c_double
    F.fk_get_int.argtypes = [tt]; F.fk_get_int.restype = C.c_longlong
    F.fk_get_str.argtypes = [tt]; F.fk_get_str.restype = C.c_char_p
    F.fk_bin_data.argtypes = [tt]; F.fk_bin_data.restype = C.c_void_p
    F.fk_bin_size.argtypes = [tt]; F.fk_bin_size.restype = C.c_size_t
    F.fk_len.argtypes = [tt]; F.fk_len.restype = C.c_uint
    F.fk_item.argtypes = [tt, C.c_uint]; F.fk_item.restype = tt
    F.fk_exception.restype = tt
    F.fk_badarg.restype = C.c_int
    F.fk_load.argtypes = [C.POINTER(Entry)]
    F.fk_call.argtypes = [C.POINTER(Entry), C.c_char_p, C.c_uint, C.POINTER(tt)]; F.fk_call.restype = tt
    F.fk_mailbox_len.restype = C.c_size_t
    F.fk_mailbox_get.argtypes = [C.c_size_t]; F.fk_mailbox_get.restype = tt
    F.fk_mailbox_wait.argtypes = [C.c_size_t, C.c_int]; F.fk_mailbox_wait.restype = C.c_int
    mods = {}
    for mod, so in shims.items():
        L = C.CDLL(so, mode=os.RTLD_LAZY)
        L.nif_init.restype = C.POINTER(Entry)
        mods[mod] = Module(F, L.nif_init())
    _cache[outdir] = (F, mods)
    return _cache[outdir]


class Module:
    def __init__(self, F, entry):
        self.F, self.entry = F, entry
        self.loaded = False

    @property
    def name(self):
        return self.entry.contents.name.decode()

    def table(self):
        e = self.entry.contents
        return [(e.funcs[i].name.decode(), e.funcs[i].arity, e.funcs[i].flags) for i in range(e.num_of_funcs)]

    def to_term(self, v):
        F = self.F
        if isinstance(v, _Term):
            return v.t
        if isinstance(v, Atom):
            return F.fk_atom(v.encode())
        if v is None:
            return F.fk_atom(b"nil")
        if isinstance(v, (bool, np.bool_)):
            return F.fk_atom(b"true" if v else b"false")
        if isinstance(v, (int, np.integer)):
            return F.fk_int(int(v))
        if isinstance(v, (float, np.floating)):
            return F.fk_double(float(v))
        if isinstance(v, np.ndarray):
            a = np.ascontiguousarray(v)
            return F.fk_binary(a.ctypes.data, a.nbytes)
        if isinstance(v, (bytes, bytearray)):
            return F.fk_binary(bytes(v), len(v))
        if isinstance(v, str):                                   # an Elixir string is a binary
            b = v.encode()
            return F.fk_binary(b, len(b))
        if isinstance(v, (list, tuple)):
            items = (C.c_ulong * max(len(v), 1))(*[self.to_term(x) for x in v])
            return F.fk_list(items, len(v))
        raise TypeError(type(v))

    def from_term(self, t):
        F = self.F
        ty = F.fk_type(t)
        if ty == T_ATOM:
            s = F.fk_get_str(t).decode()
            return {"true": True, "false": False, "nil": None}.get(s, Atom(s))
        if ty == T_DOUBLE:
            return F.fk_get_double(t)
        if ty == T_INT:
            return F.fk_get_int(t)
        if ty == T_STR:
            return F.fk_get_str(t).decode()
        if ty == T_BIN:
            return C.string_at(F.fk_bin_data(t), F.fk_bin_size(t))
        if ty in (T_LIST, T_TUPLE):
            items = [self.from_term(F.fk_item(t, i)) for i in range(F.fk_len(t))]
            return items if ty == T_LIST else tuple(items)
        if ty == T_MAP:
            n = F.fk_len(t)
            return {str(self.from_term(F.fk_item(t, i))): self.from_term(F.fk_item(t, n + i)) for i in range(n)}
        if ty == T_RES:
            return _Term(t)
        raise ValueError("invalid term %d" % t)

    def call(self, fn, *args):
        """Module.fn(args...) as the BEAM would dispatch it: by name and arity."""
        F = self.F
        if not self.loaded:
            assert F.fk_load(self.entry) == 0
            self.loaded = True
        argv = (C.c_ulong * max(len(args), 1))(*[self.to_term(a) for a in args])
        r = F.fk_call(self.entry, fn.encode(), len(args), argv)
        if F.fk_badarg() == 2:
            raise AttributeError("%s.%s/%d is undefined" % (self.name, fn, len(args)))
        if F.fk_badarg():
            raise BadArg("%s/%d" % (fn, len(args)))      # ArgumentError on a BEAM
        exc = F.fk_exception()
        if exc:
            raise Raised(self.from_term(exc))
        return self.from_term(r)


def mailbox(mod, n, timeout_ms=60000):
    """The first n messages the fake runtime's one process received through enif_send (waits for
    them; no term is touched while a sender thread may still be running)."""
    F = mod.F
    if not F.fk_mailbox_wait(n, timeout_ms):
        raise TimeoutError("%d of %d messages" % (F.fk_mailbox_len(), n))
    return [mod.from_term(F.fk_mailbox_get(i)) for i in range(n)]


class _Term:
    """An opaque term kept as is (resources)."""

    def __init__(self, t):
        self.t = t


def f64(b):
    return np.frombuffer(b, dtype=np.float64).copy()


def i32(b):
    return np.frombuffer(b, dtype=np.int32).copy()
