"""ctypes binding of libvgan_hip.so (C ABI: include/vgan_hip.h).  Loading never touches the GPU;
the library must exist -- there is no fallback."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvgan_hip.so")

_p = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_i64 = ctypes.c_int64
_u64 = ctypes.c_uint64

class FinalizeJob(ctypes.Structure):
    """struct vgan_finalize_job (include/vgan_hip.h): the step tail riding in a backward launch."""
    _fields_ = [("partial", _p), ("tiles", _p), ("colpart", _p), ("colkey", _p), ("stats", _p), ("loss", _p), ("loss_accum", _p),
                ("step_counter", _p), ("ntiles", ctypes.c_int32), ("chunks", ctypes.c_int32), ("n", ctypes.c_int32),
                ("d", ctypes.c_int32), ("weight", _f), ("accum_scale", _f), ("mode", ctypes.c_int32), ("ntiles_main", ctypes.c_int32)]


class GemmProblem(ctypes.Structure):
    """struct vgan_gemm_problem (include/vgan_hip.h)."""
    _fields_ = [("a", _p), ("b", _p), ("c", _p), ("kind", ctypes.c_int32), ("m", ctypes.c_int32), ("n", ctypes.c_int32),
                ("k", ctypes.c_int32), ("lda", ctypes.c_int32), ("ldb", ctypes.c_int32), ("ldc", ctypes.c_int32), ("splitk", ctypes.c_int32),
                ("d", _p), ("scratch", _p), ("ldd", ctypes.c_int32), ("k2", ctypes.c_int32)]


class LogitsChain(ctypes.Structure):
    """struct vgan_logits_chain (include/vgan_hip.h): the collapsed generator evaluated inside the mask / projection launch."""
    _fields_ = [("za", _p), ("At4", _p), ("ldza", ctypes.c_int32), ("ldat", ctypes.c_int32), ("e0", ctypes.c_int32), ("pad", ctypes.c_int32)]


class XXJob(ctypes.Structure):
    """struct vgan_xx_job (include/vgan_hip.h): the X-X Gram tiles riding in the mask / projection launch."""
    _fields_ = [("Dh", _p), ("Dl", _p), ("dsq", _p), ("tiles", _p), ("bw", _p), ("partial", _p), ("ldd", ctypes.c_int32),
                ("ntiles", ctypes.c_int32)]


class AdadeltaLayer(ctypes.Structure):
    """struct vgan_adadelta_layer (include/vgan_hip.h)."""
    _fields_ = [("w_packed", _p), ("off_w", _i64), ("off_b", _i64), ("ldp", ctypes.c_int32), ("out", ctypes.c_int32),
                ("inp", ctypes.c_int32), ("pad", ctypes.c_int32)]


class GroupedExtras(ctypes.Structure):
    """struct vgan_grouped_extras (include/vgan_hip.h): jobs riding in a grouped-GEMM launch."""
    _fields_ = [("copy_src", _p), ("copy_dst", _p), ("copy_count", _i64), ("adadelta", ctypes.c_int32), ("pad", ctypes.c_int32),
                ("p", _p), ("sq_avg", _p), ("acc_delta", _p), ("lr", _f), ("rho", _f), ("eps", _f), ("weight_decay", _f),
                ("grad_scale", _f), ("ld_extra", ctypes.c_int32), ("layer", AdadeltaLayer * 5), ("g_extra", _p), ("next_noise", _p),
                ("noise_rows", ctypes.c_int32), ("noise_cols", ctypes.c_int32), ("noise_ld", ctypes.c_int32),
                ("noise_ones_col", ctypes.c_int32), ("seed", _u64), ("step_counter", _p), ("fold", _p)]


GEMM_NN, GEMM_NT, GEMM_TN, GEMM_NT_NT = 0, 1, 2, 3
GEMM_MAX_GROUP = 4

# name -> (restype, argtypes); must list every function declared in include/vgan_hip.h
SIGNATURES = {
    "vgan_abi_version": (_i, []),
    "vgan_last_error": (ctypes.c_char_p, []),
    "vgan_linear_forward": (_i, [_p, _i, _i, _i64, _p, _i, _p, _p, _i, _i, _i, _i, _p]),
    "vgan_linear_backward_input": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _p]),
    "vgan_linear_backward_params": (_i, [_p, _i, _p, _i, _i, _i64, _p, _i, _p, _i, _i, _i, _i, _i64, _p]),
    "vgan_linear_backward_params_xx_supported": (_i, [_i, _i, _i]),
    "vgan_linear_backward_params_xx": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _p, _p]),
    "vgan_reduce_slabs": (_i, [_p, _i64, _i, _p, _i64, _p]),
    "vgan_mask_project_forward": (_i, [_p, _i, _p, _i, _p, _p, _i, _i, _i, _p, _p, _p, _p, _i, _p, _p, _i, _i, _p, _i, _p, _p]),
    "vgan_col_mean": (_i, [_p, _i, _i, _i, _p, _p]),
    "vgan_gather_rows": (_i, [_p, _i, _p, _p, _i, _i, _i, _p, _i, _p, _i, _i, _p]),
    "vgan_gather_rows_split": (_i, [_p, _i, _p, _p, _i, _i, _i, _p, _p, _i, _p, _i, _p, _p, _i, _i, _i, _p]),
    "vgan_mask_backward": (_i, [_p, _i, _i, _i64, _p, _i, _p, _f, _i, _p, _i, _i, _i, _p]),
    "vgan_colmax_partial": (_i, [_p, _i, _i, _i, _p, _i, _i, _p]),
    "vgan_mmd_finalize": (_i, [_p, _p, _i, _p, _i, _p, _i, _i, _f, _p, _p, _p, _f, _p, _p]),
    "vgan_colmax_chunks": (_i, [_i]),
    "vgan_colmax": (_i, [_p, _i, _i, _i, _p, _p, _i, _i, _p]),
    "vgan_mask_from_softmax": (_i, [_p, _i, _p, _i, _i, _i, _p]),
    "vgan_upper_softmax_forward": (_i, [_p, _i, _p, _p, _i, _i, _p]),
    "vgan_mmd_build_tiles": (_i, [_i, _i, _i, _i, _i, _p, _i]),
    "vgan_mmd_order_tiles": (_i, [_p, _i, _i]),
    "vgan_mmd_gram": (_i, [_p, _i, _p, _i, _i, _p, _p, _i, _i, _p, _i, _i, _p, _p]),
    "vgan_mmd_gram_general": (_i, [_p, _i, _p, _i, _i, _p, _p, _i, _p, _i, _p, _i, _i, _p, _p]),
    "vgan_mmd_gram_colmax": (_i, [_p, _i, _p, _i, _i, _p, _p, _i, _p, _i, _i, _p, _p, _i, _i, _i, _p, _i, _i, _p]),
    "vgan_mmd_reduce": (_i, [_p, _p, _i, _p, _i, _p]),
    "vgan_mmd_set_bandwidth": (_i, [_p, _i, _p, _p]),
    "vgan_mmd_loss": (_i, [_p, _p, _i, _i, _f, _p, _p, _f, _p, _p]),
    "vgan_mmd_backward": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _p, _i, _p, _p, _i, _i, _i64, _p, _p]),
    "vgan_mmd_bf3_prepare": (_i, [_p, _i, _i, _i, _p, _p, _i, _p, _p, _i, _p]),
    "vgan_mmd_gram_bf3": (_i, [_p, _p, _i, _p, _i, _p, _p, _i, _i, _p, _p, _i, _i, _p, _p, _i, _i, _i, _p, _i, _i, _p, _i64, _p, _i, _p]),
    "vgan_mmd_gram_bf3_tail_ws_bytes": (_i64, []),
    "vgan_mmd_backward_bf3": (_i, [_p, _p, _i, _p, _p, _i, _i, _p, _i, _i, _i, _i, _p, _i, _p, _p, _i, _i, _i64, _i, _p, _p]),
    "vgan_mmd_backward_bf3_tile": (_i, [_i, _i, _i, _i]),
    "vgan_mmd_backward_bf3_rm": (_i, [_p, _p, _i, _i, _p, _p, _i, _i, _p, _i, _i, _i, _i, _p, _i, _p, _p, _i, _i, _i64, _i, _p, _p, _i, _p]),
    "vgan_mmd_backward_bf3_rm_xx": (_i, [_p, _p, _i, _i, _p, _p, _i, _i, _p, _i, _i, _i, _i, _p, _i, _p, _p, _i, _i, _i64, _p, _p, _p]),
    "vgan_row_sqnorm": (_i, [_p, _i, _p, _i, _i, _p]),
    "vgan_adadelta_step": (_i, [_p, _p, _i, _i64, _p, _p, _i64, _f, _f, _f, _f, _f, _p]),
    "vgan_noise_normal": (_i, [_p, _i, _i, _i, _i, _u64, _p, _u64, _p]),
    "vgan_homogeneous_pack": (_i, [_p, _i, _i, _i, _p]),
    "vgan_adadelta_step_packed": (_i, [_p, _p, _p, _p, _p, _p, _i64, _f, _f, _f, _f, _f, _p, _i, _i, _i, _i, _u64, _p, _p]),
    "vgan_gemm_grouped": (_i, [_p, _i, _p]),
    "vgan_gemm_grouped_ex": (_i, [_p, _i, _p, _p]),
    "vgan_mask_project_forward_bf3": (_i, [_p, _i, _p, _i, _p, _p, _i, _i, _p, _p, _i, _p, _p, _p, _i, _p, _p, _i, _i, _i, _p, _i, _p, _p, _p]),
    "vgan_mse_grad": (_i, [_p, _i, _p, _i, _i, _i, _f, _p, _p, _i, _p]),
    "vgan_sum_f64": (_i, [_p, _i, ctypes.c_double, _p, _i, _p]),
    "vgan_rbf_kernel_matrix": (_i, [_p, _i, _i, _i, _p, _f, _p, _i, _p]),
    "vgan_rbf_multi_kernel_matrix": (_i, [_p, _i, _i, _i, _p, _p, _p, _i, _p, _i, _p, _i, _p]),
    "vgan_rows_dot": (_i, [_p, _i, _p, _i, _p, _i, _i, _p]),
    "vgan_shuffle_epoch": (_i, [_p, _i64, _i64, _u64, _u64, _p]),
    "vgan_shuffle_index": (_i64, [_i64, _i64, _u64, _u64]),
    "vgan_mask_unique": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _p]),
    "vgan_dp_unique_id": (_i, [_p]),
    "vgan_dp_comm_create": (_i, [_p, _i, _p, _i]),
    "vgan_dp_allreduce_sum": (_i, [_p, _p, _i64, _p]),
    "vgan_dp_allgather": (_i, [_p, _p, _i64, _p]),
    "vgan_dp_comm_destroy": (_i, [_p]),
    "vgan_mse": (_i, [_p, _i, _p, _i, _i, _i, _f, _p, _i, _p]),
}

ABI_VERSION = 6
_lib = None


class VganHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle.  Raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VganHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C v-gan_amd/csrc`).  vgan_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.vgan_abi_version() != ABI_VERSION:
        raise VganHipError(f"ABI mismatch: library {lib.vgan_abi_version()} vs binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().vgan_last_error().decode(errors="replace")
        raise VganHipError(f"{what} failed (code {rc}): {msg}")


def build_tiles(n, grad_mode, rank=0, world=1, tile=64):
    """Host-side tile table of the Gram kernel as a flat list of int32 (8 per tile)."""
    lib = load()
    cnt = lib.vgan_mmd_build_tiles(n, grad_mode, rank, world, tile, None, 0)
    if cnt < 0:
        raise VganHipError("vgan_mmd_build_tiles: " + lib.vgan_last_error().decode())
    buf = (ctypes.c_int32 * (cnt * 8))()
    got = lib.vgan_mmd_build_tiles(n, grad_mode, rank, world, tile, ctypes.cast(buf, ctypes.c_void_p), cnt)
    if got != cnt:
        raise VganHipError("vgan_mmd_build_tiles: inconsistent tile count")
    return list(buf), cnt


def split_tiles(table, tile=64, yy_last=False):
    """Splits a tile table (torch int32 [count, 8], host) into two parts, each re-ordered for the XCDs:
    default   (tiles of the XY / YY blocks, tiles of the XX block) -- the XX tiles only feed the reported loss (sums, no
              gradient weights), so a step may run them in another launch;
    yy_last   (tiles of the XY / XX blocks, tiles of the YY block) -- the sharded front of a data-parallel step: the first
              part reads no other rank's Y rows and runs beside their all-gather."""
    import torch
    lib = load()
    slot = table[:, 4] & 3
    first = (slot != 2) if yy_last else (slot != 0)
    parts = []
    for keep in (first, ~first):
        sub = table[keep].contiguous()
        if sub.shape[0] > 1:
            buf = (ctypes.c_int32 * sub.numel())(*sub.reshape(-1).tolist())
            if lib.vgan_mmd_order_tiles(ctypes.cast(buf, ctypes.c_void_p), sub.shape[0], tile) != 0:
                raise VganHipError("vgan_mmd_order_tiles: " + lib.vgan_last_error().decode())
            sub = torch.tensor(list(buf), dtype=torch.int32).view(-1, 8)
        parts.append(sub)
    return parts
