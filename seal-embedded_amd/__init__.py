"""seal-embedded_amd: Python (ctypes) binding of libseal_embedded_amd.so.

This is plumbing for tests and bench.py: PyTorch supplies device memory and streams, every
compute call goes straight through the C ABI declared in include/seal_embedded_amd.h.
There is no Python or CPU implementation of the path here -- if the HIP library is missing or
no GPU is present, construction fails loudly.

The directory name contains a hyphen, so load it through `__graft_entry__.load_package()` (or
importlib) under the module name `seal_embedded_amd`.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libseal_embedded_amd.so")
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include", "seal_embedded_amd.h")

STAGES = ("cbd", "uniform", "ternary", "encode_encrypt", "encode_rns", "ntt_fuse")

SE_SUCCESS = 0


class SealEmbeddedAmdError(RuntimeError):
    pass


TESTHOOKS_LIB_DIR = os.path.join(HERE, "lib", "testhooks")   # the -DSEAMD_TEST_HOOKS build (tests only)


def build_library(jobs=8, verbose=False):
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU), and beside it the
    test build (se_api.cpp under -DSEAMD_TEST_HOOKS: fault injection) the GPU tests link against."""
    cmd = ["make", "-C", CSRC, f"-j{jobs}", "all", "testhooks"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None

# every symbol include/seal_embedded_amd.h declares (checked by tests/test_cabi.py)
EXPORTED_SYMBOLS = (
    "se_setup_custom", "se_setup", "se_setup_default", "se_encrypt_seeded", "se_encrypt",
    "se_cleanup", "se_encrypt_batch",
    "se_amd_create", "se_amd_destroy", "se_amd_degree", "se_amd_nprimes", "se_amd_scale",
    "se_amd_moduli", "se_amd_index_map", "se_amd_set_secret_key", "se_amd_set_public_key",
    "se_amd_load_keys_from_dir", "se_amd_gen_public_key", "se_amd_gen_keys_batch", "se_amd_encrypt_sym_device", "se_amd_encrypt_asym_device", "se_amd_encrypt_sym_seeded_device",
    "se_amd_expand_c1_device",
    "se_amd_encode_ntt_device", "se_amd_encrypt_sym_host", "se_amd_encrypt_asym_host",
    "se_amd_encode_device", "se_amd_ntt_device", "se_amd_intt_device", "se_amd_decrypt_decode_device", "se_amd_prng_blocks_device",
    "se_amd_sample_uniform_device", "se_amd_sample_ternary_device", "se_amd_sample_cbd_device",
    "se_amd_pack_ternary_host", "se_amd_word_ops_device", "se_amd_pack_seal_ciphertext_host", "se_amd_format_poly_text",
    "se_amd_format_values_text", "se_amd_write_ciphertext_text", "se_amd_save_secret_key_file",
    "se_amd_save_public_key_files", "se_amd_set_profiling", "se_amd_stage_ms",
    "se_amd_group_create", "se_amd_group_destroy", "se_amd_group_size", "se_amd_group_ctx", "se_amd_group_device",
    "se_amd_group_partition", "se_amd_group_set_secret_key", "se_amd_group_set_public_key", "se_amd_group_reserve",
    "se_amd_encrypt_sym_multi_device", "se_amd_encrypt_asym_multi_device", "se_amd_encode_ntt_multi_device",
    "se_amd_set_reject_list_capacity", "se_amd_set_speculation_capacity", "se_amd_set_host_chunk", "se_amd_host_tables", "se_amd_ifft_table_sha256", "se_amd_reserve", "se_amd_set_debug_flags", "se_amd_set_pipeline", "se_amd_set_asym_chunks", "se_amd_last_error", "se_amd_version",
)


def lib():
    """Load the shared library (never builds implicitly; never falls back)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SealEmbeddedAmdError(
            f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, sz, i32, u32 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint32
    L.se_amd_last_error.restype = C.c_char_p
    L.se_amd_version.restype = C.c_char_p
    L.se_amd_create.argtypes = [C.POINTER(vp), sz, sz, i32]
    L.se_amd_destroy.argtypes = [vp]
    L.se_amd_destroy.restype = None
    L.se_amd_degree.argtypes = [vp]
    L.se_amd_degree.restype = sz
    L.se_amd_nprimes.argtypes = [vp]
    L.se_amd_nprimes.restype = sz
    L.se_amd_scale.argtypes = [vp]
    L.se_amd_scale.restype = C.c_double
    L.se_amd_moduli.argtypes = [vp, vp]
    L.se_amd_index_map.argtypes = [vp, vp]
    L.se_amd_set_secret_key.argtypes = [vp, vp]
    L.se_amd_set_public_key.argtypes = [vp, vp, vp]
    L.se_amd_load_keys_from_dir.argtypes = [vp, C.c_char_p, i32]
    L.se_amd_gen_public_key.argtypes = [vp, vp, vp, vp, vp, vp]
    L.se_amd_gen_keys_batch.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, vp]
    L.se_amd_encrypt_sym_device.argtypes = [vp, vp, sz, vp, vp, vp, vp, vp, vp, vp, vp]
    L.se_amd_encrypt_asym_device.argtypes = [vp, vp, sz, vp, vp, vp, vp, vp, vp, vp]
    L.se_amd_encrypt_sym_seeded_device.argtypes = [vp, vp, sz, vp, vp, vp, vp, vp]
    L.se_amd_expand_c1_device.argtypes = [vp, vp, sz, vp, vp]
    L.se_amd_encode_ntt_device.argtypes = [vp, vp, sz, vp, vp, vp, vp]
    L.se_amd_encrypt_sym_host.argtypes = [vp, vp, sz, vp, vp, vp, vp, vp, vp, vp]
    L.se_amd_encrypt_asym_host.argtypes = [vp, vp, sz, vp, vp, vp, vp, vp, vp]
    L.se_amd_encode_device.argtypes = [vp, vp, sz, vp, vp, vp]
    L.se_amd_ntt_device.argtypes = [vp, sz, vp, sz, vp]
    L.se_amd_intt_device.argtypes = [vp, sz, vp, sz, vp]
    L.se_amd_decrypt_decode_device.argtypes = [vp, vp, vp, sz, sz, vp, vp, vp, vp]
    L.se_amd_prng_blocks_device.argtypes = [vp, vp, vp, vp, sz, sz, vp]
    L.se_amd_sample_uniform_device.argtypes = [vp, vp, vp, sz, vp, vp, vp]
    L.se_amd_sample_ternary_device.argtypes = [vp, vp, sz, vp, vp, vp]
    L.se_amd_sample_cbd_device.argtypes = [vp, vp, vp, sz, sz, vp, vp]
    L.se_amd_word_ops_device.argtypes = [vp, sz, i32, vp, vp, vp, vp, sz, vp]
    L.se_amd_pack_ternary_host.argtypes = [vp, sz, vp]
    L.se_amd_pack_ternary_host.restype = None
    L.se_amd_pack_seal_ciphertext_host.argtypes = [vp, vp, sz, sz, vp]
    L.se_amd_pack_seal_ciphertext_host.restype = None
    L.se_amd_format_poly_text.argtypes = [C.c_char_p, vp, sz, vp, sz]
    L.se_amd_format_poly_text.restype = sz
    L.se_amd_format_values_text.argtypes = [C.c_char_p, vp, sz, vp, sz]
    L.se_amd_format_values_text.restype = sz
    L.se_amd_write_ciphertext_text.argtypes = [C.c_char_p, i32, vp, sz, vp, vp, sz, sz]
    L.se_amd_save_secret_key_file.argtypes = [C.c_char_p, sz, vp]
    L.se_amd_save_public_key_files.argtypes = [C.c_char_p, sz, sz, vp, vp, vp]
    L.se_amd_set_profiling.argtypes = [vp, i32]
    L.se_amd_stage_ms.argtypes = [vp, vp, vp, i32]
    L.se_amd_set_reject_list_capacity.argtypes = [vp, u32]
    L.se_amd_set_speculation_capacity.argtypes = [vp, u32]
    L.se_amd_set_host_chunk.argtypes = [vp, sz]
    L.se_amd_host_tables.argtypes = [sz, sz, vp, vp, vp, vp, vp, vp, vp]
    L.se_amd_ifft_table_sha256.argtypes = [vp, C.c_char_p]
    L.se_amd_reserve.argtypes = [vp, sz]
    L.se_amd_set_debug_flags.argtypes = [vp, u32]
    L.se_amd_set_pipeline.argtypes = [vp, i32, i32]
    L.se_amd_set_asym_chunks.argtypes = [vp, sz]
    _lib = L
    return L


def _check(rc, what):
    if rc < 0:
        msg = lib().se_amd_last_error().decode(errors="replace")
        raise SealEmbeddedAmdError(f"{what} failed with code {rc}: {msg}")
    return rc


def _ptr(t):
    """Device/host pointer of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        assert t.is_contiguous()
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)


def _stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def host_tables(n, nprimes):
    """Setup-time tables computed on the host (no GPU needed): dict of numpy arrays."""
    import numpy as np
    L = lib()
    q = np.zeros(nprimes, np.uint32); cr = np.zeros((nprimes, 2), np.uint32)
    scale = C.c_double(0)
    imap = np.zeros(n, np.uint16); w = np.zeros((n, 2), np.float64)
    rw = np.zeros((nprimes, n, 2), np.uint32); irw = np.zeros_like(rw)
    _check(L.se_amd_host_tables(n, nprimes, _ptr(q), _ptr(cr), C.c_void_p(C.addressof(scale)), _ptr(imap), _ptr(w),
                                _ptr(rw), _ptr(irw)), "se_amd_host_tables")
    return dict(q=q, const_ratio=cr, scale=scale.value, index_map=imap, ifft_w=w, ntt_rw=rw, intt_rw=irw)


class Group:
    """One context per device of a node (se_amd_group): device-resident multi-GPU calls through the C ABI.
    `blocks` arguments are lists with one torch tensor per member, each on that member's device."""

    def __init__(self, n, nprimes, devices=None):
        self.L = lib()
        L = self.L
        L.se_amd_group_size.restype = C.c_size_t
        L.se_amd_group_ctx.restype = C.c_void_p
        L.se_amd_group_ctx.argtypes = [C.c_void_p, C.c_size_t]
        L.se_amd_group_destroy.restype = None
        L.se_amd_group_destroy.argtypes = [C.c_void_p]
        h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices) if devices else None
        _check(L.se_amd_group_create(C.byref(h), C.c_size_t(n), C.c_size_t(nprimes), arr,
                                     C.c_size_t(len(devices) if devices else 0)), "se_amd_group_create")
        self.h, self.n, self.np = h, n, nprimes
        self.size = int(L.se_amd_group_size(h))
        self.devices = [int(L.se_amd_group_device(h, C.c_size_t(i))) for i in range(self.size)]

    def close(self):
        if getattr(self, "h", None):
            self.L.se_amd_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def partition(self, B):
        first = (C.c_size_t * self.size)()
        count = (C.c_size_t * self.size)()
        _check(self.L.se_amd_group_partition(self.h, C.c_size_t(B), first, count), "se_amd_group_partition")
        return list(first), list(count)

    def set_secret_key(self, sk_packed):
        import numpy as np
        sk = np.ascontiguousarray(sk_packed, dtype=np.uint8)
        _check(self.L.se_amd_group_set_secret_key(self.h, _ptr(sk)), "se_amd_group_set_secret_key")

    def set_public_key(self, pk0, pk1):
        import numpy as np
        pk0 = np.ascontiguousarray(pk0, dtype=np.uint32)
        pk1 = np.ascontiguousarray(pk1, dtype=np.uint32)
        _check(self.L.se_amd_group_set_public_key(self.h, _ptr(pk0), _ptr(pk1)), "se_amd_group_set_public_key")

    def reserve(self, B):
        _check(self.L.se_amd_group_reserve(self.h, C.c_size_t(B)), "se_amd_group_reserve")

    def _arr(self, blocks):
        if blocks is None:
            return None
        assert len(blocks) == self.size
        return (C.c_void_p * self.size)(*[None if t is None else t.data_ptr() for t in blocks])

    def encrypt_sym(self, B, values, share_seeds, seeds, c0, c1=None, status=None, gather_root=-1, c0_all=None,
                    c1_all=None):
        _check(self.L.se_amd_encrypt_sym_multi_device(
            self.h, C.c_size_t(B), self._arr(values), self._arr(share_seeds), self._arr(seeds), self._arr(c0),
            self._arr(c1), self._arr(status), C.c_int(gather_root), _ptr(c0_all), _ptr(c1_all)),
            "se_amd_encrypt_sym_multi_device")

    def encrypt_asym(self, B, values, seeds, c0, c1, status=None, gather_root=-1, c0_all=None, c1_all=None):
        _check(self.L.se_amd_encrypt_asym_multi_device(
            self.h, C.c_size_t(B), self._arr(values), self._arr(seeds), self._arr(c0), self._arr(c1),
            self._arr(status), C.c_int(gather_root), _ptr(c0_all), _ptr(c1_all)),
            "se_amd_encrypt_asym_multi_device")

    def encode_ntt(self, B, values, out, status=None, gather_root=-1, out_all=None):
        _check(self.L.se_amd_encode_ntt_multi_device(
            self.h, C.c_size_t(B), self._arr(values), self._arr(out), self._arr(status), C.c_int(gather_root),
            _ptr(out_all)), "se_amd_encode_ntt_multi_device")


class Context:
    """One parameter set on one GPU (se_amd_ctx).  All tensors are torch CUDA tensors."""

    def __init__(self, n, nprimes, device=0):
        self.L = lib()
        h = C.c_void_p()
        _check(self.L.se_amd_create(C.byref(h), n, nprimes, device), "se_amd_create")
        self.h = h
        self.n, self.np, self.device = n, nprimes, device

    def close(self):
        if getattr(self, "h", None):
            self.L.se_amd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters / keys
    def scale(self):
        return float(self.L.se_amd_scale(self.h))

    def moduli(self):
        import numpy as np
        q = np.zeros(self.np, dtype=np.uint32)
        _check(self.L.se_amd_moduli(self.h, _ptr(q)), "se_amd_moduli")
        return [int(x) for x in q]

    def index_map(self):
        import numpy as np
        m = np.zeros(self.n, dtype=np.uint16)
        _check(self.L.se_amd_index_map(self.h, _ptr(m)), "se_amd_index_map")
        return m

    def ifft_table_sha256(self):
        """SHA-256 of the IFFT root table as the DEVICE holds it (SURVEY trap T8)."""
        buf = C.create_string_buffer(65)
        _check(self.L.se_amd_ifft_table_sha256(self.h, buf), "se_amd_ifft_table_sha256")
        return buf.value.decode()

    def set_secret_key(self, sk_packed):
        import numpy as np
        sk = np.ascontiguousarray(sk_packed, dtype=np.uint8)
        assert sk.size == self.n // 4
        _check(self.L.se_amd_set_secret_key(self.h, _ptr(sk)), "se_amd_set_secret_key")

    def set_public_key(self, pk0, pk1):
        import numpy as np
        pk0 = np.ascontiguousarray(pk0, dtype=np.uint32)
        pk1 = np.ascontiguousarray(pk1, dtype=np.uint32)
        assert pk0.size == self.np * self.n == pk1.size
        _check(self.L.se_amd_set_public_key(self.h, _ptr(pk0), _ptr(pk1)), "se_amd_set_public_key")

    def gen_public_key(self, sk_packed, pk_seed, ep_seed):
        import numpy as np
        sk = np.ascontiguousarray(sk_packed, dtype=np.uint8)
        s1 = np.frombuffer(bytes(pk_seed), dtype=np.uint8).copy()
        s2 = np.frombuffer(bytes(ep_seed), dtype=np.uint8).copy()
        pk0 = np.zeros((self.np, self.n), dtype=np.uint32)
        pk1 = np.zeros_like(pk0)
        _check(self.L.se_amd_gen_public_key(self.h, _ptr(sk), _ptr(s1), _ptr(s2), _ptr(pk0),
                                            _ptr(pk1)), "se_amd_gen_public_key")
        return pk0, pk1

    def gen_keys_batch(self, pk_seeds, ep_seeds, sk_seeds=None, sk_in=None):
        """K key pairs in one launch chain: (sk [K][n/4] uint8, pk0, pk1 [K][np][n] uint32)."""
        import numpy as np
        pks = np.ascontiguousarray(pk_seeds, dtype=np.uint8).reshape(-1, 64)
        K = pks.shape[0]
        eps = np.ascontiguousarray(ep_seeds, dtype=np.uint8).reshape(K, 64)
        sks = None if sk_seeds is None else np.ascontiguousarray(sk_seeds, dtype=np.uint8).reshape(K, 64)
        ski = None if sk_in is None else np.ascontiguousarray(sk_in, dtype=np.uint8).reshape(K, self.n // 4)
        sk = np.zeros((K, self.n // 4), dtype=np.uint8)
        pk0 = np.zeros((K, self.np, self.n), dtype=np.uint32)
        pk1 = np.zeros_like(pk0)
        _check(self.L.se_amd_gen_keys_batch(self.h, K, _ptr(ski), _ptr(sks), _ptr(pks), _ptr(eps), _ptr(sk),
                                            _ptr(pk0), _ptr(pk1)), "se_amd_gen_keys_batch")
        return sk, pk0, pk1

    def load_keys_from_dir(self, path, want_pk=False):
        _check(self.L.se_amd_load_keys_from_dir(self.h, path.encode(), 1 if want_pk else 0),
               "se_amd_load_keys_from_dir")

    # ---- whole path (device tensors, async on torch's current stream)
    def encrypt_sym(self, values, share_seeds, seeds, c0, c1, ntt_pte=None, pte=None, status=None):
        B = values.shape[0]
        _check(self.L.se_amd_encrypt_sym_device(self.h, _ptr(values), B, _ptr(share_seeds),
                                                _ptr(seeds), _ptr(c0), _ptr(c1), _ptr(ntt_pte),
                                                _ptr(pte), _ptr(status), _stream_ptr()),
               "se_amd_encrypt_sym_device")

    def encrypt_sym_seeded(self, values, share_seeds, seeds, c0, status=None):
        _check(self.L.se_amd_encrypt_sym_seeded_device(self.h, _ptr(values), values.shape[0],
                                                       _ptr(share_seeds), _ptr(seeds), _ptr(c0),
                                                       _ptr(status), _stream_ptr()),
               "se_amd_encrypt_sym_seeded_device")

    def expand_c1(self, share_seeds, c1):
        _check(self.L.se_amd_expand_c1_device(self.h, _ptr(share_seeds), share_seeds.shape[0],
                                              _ptr(c1), _stream_ptr()), "se_amd_expand_c1_device")

    def encrypt_asym(self, values, seeds, c0, c1, ntt_pte=None, pte=None, status=None):
        B = values.shape[0]
        _check(self.L.se_amd_encrypt_asym_device(self.h, _ptr(values), B, _ptr(seeds), _ptr(c0),
                                                 _ptr(c1), _ptr(ntt_pte), _ptr(pte), _ptr(status),
                                                 _stream_ptr()), "se_amd_encrypt_asym_device")

    def encode_ntt(self, values, out, pte=None, status=None):
        B = values.shape[0]
        _check(self.L.se_amd_encode_ntt_device(self.h, _ptr(values), B, _ptr(out), _ptr(pte),
                                               _ptr(status), _stream_ptr()),
               "se_amd_encode_ntt_device")

    # ---- stage level
    def encode(self, values, out, status=None):
        _check(self.L.se_amd_encode_device(self.h, _ptr(values), values.shape[0], _ptr(out),
                                           _ptr(status), _stream_ptr()), "se_amd_encode_device")

    def ntt(self, prime, polys):
        count = polys.numel() // self.n
        _check(self.L.se_amd_ntt_device(self.h, prime, _ptr(polys), count, _stream_ptr()),
               "se_amd_ntt_device")

    def intt(self, prime, polys):
        count = polys.numel() // self.n
        _check(self.L.se_amd_intt_device(self.h, prime, _ptr(polys), count, _stream_ptr()),
               "se_amd_intt_device")

    def decrypt_decode(self, c0, c1, prime, dec_ntt=None, pt=None, values=None):
        B = c0.shape[0]
        _check(self.L.se_amd_decrypt_decode_device(self.h, _ptr(c0), _ptr(c1), B, prime,
                                                   _ptr(dec_ntt), _ptr(pt), _ptr(values),
                                                   _stream_ptr()), "se_amd_decrypt_decode_device")

    def prng_blocks(self, seeds, ctrs, out, outlen):
        _check(self.L.se_amd_prng_blocks_device(self.h, _ptr(seeds), _ptr(ctrs), _ptr(out), outlen,
                                                seeds.shape[0], _stream_ptr()),
               "se_amd_prng_blocks_device")

    def sample_uniform(self, seeds, out, ctr_in=None, ctr_out=None):
        _check(self.L.se_amd_sample_uniform_device(self.h, _ptr(seeds), _ptr(ctr_in),
                                                   seeds.shape[0], _ptr(out), _ptr(ctr_out),
                                                   _stream_ptr()), "se_amd_sample_uniform_device")

    def sample_ternary(self, seeds, codes, ctr_out=None):
        _check(self.L.se_amd_sample_ternary_device(self.h, _ptr(seeds), seeds.shape[0],
                                                   _ptr(codes), _ptr(ctr_out), _stream_ptr()),
               "se_amd_sample_ternary_device")

    def sample_cbd(self, seeds, out, blocks_per_ct, ctr_base=None):
        _check(self.L.se_amd_sample_cbd_device(self.h, _ptr(seeds), _ptr(ctr_base), seeds.shape[0],
                                               blocks_per_ct, _ptr(out), _stream_ptr()),
               "se_amd_sample_cbd_device")

    def word_ops(self, prime, op, a, b=None, c=None):
        """Device word arithmetic KAT hook: uint64 numpy operands in, uint32 numpy out."""
        import numpy as np
        import torch
        dev = torch.device("cuda", self.device)
        t = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x, dtype=np.uint64).view(np.int64)).to(dev)
        ta, tb, tc = t(a), t(b), t(c)
        out = torch.zeros(ta.numel(), dtype=torch.int32, device=dev)
        _check(self.L.se_amd_word_ops_device(self.h, prime, op, _ptr(ta), _ptr(tb), _ptr(tc), _ptr(out),
                                             ta.numel(), _stream_ptr()), "se_amd_word_ops_device")
        torch.cuda.synchronize()
        return out.cpu().numpy().view(np.uint32)

    def pack_ternary(self, codes_np):
        import numpy as np
        codes = np.ascontiguousarray(codes_np, dtype=np.int8)
        out = np.zeros(codes.size // 4, dtype=np.uint8)
        self.L.se_amd_pack_ternary_host(_ptr(codes), codes.size, _ptr(out))
        return out

    # ---- host-pointer wrappers (numpy in / numpy out)
    def _host_out(self, out, B):
        """(c0, c1) host arrays: fresh, or the caller's (e.g. pinned) uint32[B][np][n] buffers."""
        import numpy as np
        if out is None:
            c0 = np.zeros((B, self.np, self.n), dtype=np.uint32)
            return c0, np.zeros_like(c0)
        c0, c1 = out
        for a in (c0, c1):
            if a.dtype != np.uint32 or a.shape != (B, self.np, self.n) or not a.flags.c_contiguous:
                raise ValueError("out buffers must be C-contiguous uint32[B][np][n]")
        return c0, c1

    def encrypt_sym_host(self, values, share_seeds, seeds, want_extra=False, out=None,
                         seed_compressed=False):
        import numpy as np
        v = np.ascontiguousarray(values, dtype=np.float32).reshape(-1, self.n // 2)
        B = v.shape[0]
        ss = np.ascontiguousarray(share_seeds, dtype=np.uint8).reshape(B, 64)
        sd = np.ascontiguousarray(seeds, dtype=np.uint8).reshape(B, 64)
        c0, c1 = self._host_out(out, B)
        if seed_compressed:
            c1 = None                  # only c0 crosses PCIe; c1 = expand_c1(share_seeds)
        ntt_pte = np.zeros_like(c0) if want_extra else None
        pte = np.zeros((B, self.n), dtype=np.int64) if want_extra else None
        status = np.zeros(B, dtype=np.uint8)
        rc = _check(self.L.se_amd_encrypt_sym_host(self.h, _ptr(v), B, _ptr(ss), _ptr(sd),
                                                   _ptr(c0), _ptr(c1), _ptr(ntt_pte), _ptr(pte),
                                                   _ptr(status)), "se_amd_encrypt_sym_host")
        return dict(failed=rc, c0=c0, c1=c1, ntt_pte=ntt_pte, pte=pte, status=status)

    def encrypt_asym_host(self, values, seeds, want_extra=False, out=None):
        import numpy as np
        v = np.ascontiguousarray(values, dtype=np.float32).reshape(-1, self.n // 2)
        B = v.shape[0]
        sd = np.ascontiguousarray(seeds, dtype=np.uint8).reshape(B, 64)
        c0, c1 = self._host_out(out, B)
        ntt_pte = np.zeros_like(c0) if want_extra else None
        pte = np.zeros((B, self.n), dtype=np.int64) if want_extra else None
        status = np.zeros(B, dtype=np.uint8)
        rc = _check(self.L.se_amd_encrypt_asym_host(self.h, _ptr(v), B, _ptr(sd), _ptr(c0),
                                                    _ptr(c1), _ptr(ntt_pte), _ptr(pte),
                                                    _ptr(status)), "se_amd_encrypt_asym_host")
        return dict(failed=rc, c0=c0, c1=c1, ntt_pte=ntt_pte, pte=pte, status=status)

    # ---- profiling
    def set_profiling(self, on=True):
        _check(self.L.se_amd_set_profiling(self.h, 1 if on else 0), "se_amd_set_profiling")

    def stage_ms(self, reset=True):
        """{stage: (total_ms, launches)} measured with HIP events on the launch stream."""
        ms = (C.c_float * len(STAGES))()
        cnt = (C.c_uint64 * len(STAGES))()
        _check(self.L.se_amd_stage_ms(self.h, ms, cnt, 1 if reset else 0), "se_amd_stage_ms")
        return {s: (float(ms[i]), int(cnt[i])) for i, s in enumerate(STAGES)}

    def set_pipeline(self, overlap=True, split=True):
        _check(self.L.se_amd_set_pipeline(self.h, int(overlap), int(split)), "se_amd_set_pipeline")

    def set_asym_chunks(self, chunks):
        _check(self.L.se_amd_set_asym_chunks(self.h, chunks), "se_amd_set_asym_chunks")

    def set_debug_flags(self, flags):
        _check(self.L.se_amd_set_debug_flags(self.h, flags), "se_amd_set_debug_flags")

    def set_speculation_capacity(self, cap):
        _check(self.L.se_amd_set_speculation_capacity(self.h, cap),
               "se_amd_set_speculation_capacity")

    def set_host_chunk(self, cts):
        _check(self.L.se_amd_set_host_chunk(self.h, cts), "se_amd_set_host_chunk")

    def reserve(self, B):
        _check(self.L.se_amd_reserve(self.h, B), "se_amd_reserve")

    def set_reject_list_capacity(self, cap):
        _check(self.L.se_amd_set_reject_list_capacity(self.h, cap),
               "se_amd_set_reject_list_capacity")
