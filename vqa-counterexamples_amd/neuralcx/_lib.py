"""ctypes binding of libneuralcx_hip.so (C ABI: include/neuralcx.h).

This is the stub a maintainer of the reference would add: plain pointers and
sizes in, status code out.  There is NO fallback: if the HIP library is missing or fails to load,
importing a symbol raises -- the product path never routes through a CPU implementation.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NCX_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libneuralcx_hip.so")   # (NCX_LIB: A/B builds, tools/)

NCX_F_V_MULT, NCX_F_V_DIST, NCX_F_V_RANK, NCX_F_A_EMB = 1, 2, 4, 8
NCX_F_ALL = 15
NCX_F_REUSE_GT = 32   # evaluation: Gt in the workspace is still valid (same weights)
NCX_F_FUSED_TAIL = 64  # training: out layer + loss / Recall + head of the backward in one pass (ncx_train_tail)
NCX_F_X6 = 128         # NOT the default: the balanced TN weight-gradient launch on the bf16 matrix path with three-plane (fp32-grade) operands
NCX_F_BF16 = 16       # BASELINE configs[4]: bf16 operands for the two dominant GEMMs (include/neuralcx.h)

EXPORTS = ("ncx_input_size", "ncx_workspace_bytes", "ncx_forward", "ncx_forward_phase", "ncx_loss_rank", "ncx_backward", "ncx_backward_phase",
           "ncx_adam_step", "ncx_version", "ncx_profile_begin", "ncx_profile_end", "ncx_plan_query",
           "ncx_vqa_workspace_bytes", "ncx_vqa_forward", "ncx_knn_workspace_bytes", "ncx_knn", "ncx_ws_region", "ncx_wgmap_check",
           "ncx_comm_unique_id", "ncx_comm_create", "ncx_comm_destroy", "ncx_allreduce", "ncx_train_tail", "ncx_profile_stamps")


class NcxDims(C.Structure):
    _fields_ = [("B", C.c_int32), ("K", C.c_int32), ("dv", C.c_int32), ("dq", C.c_int32),
                ("dz", C.c_int32), ("da", C.c_int32), ("A", C.c_int32), ("H", C.c_int32),
                ("L", C.c_int32), ("n_img", C.c_int32), ("flags", C.c_uint32),
                ("training", C.c_int32), ("drop_p", C.c_float), ("loss_scale", C.c_float),
                ("seed", C.c_uint64)]


class NcxInputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("feats", "img_idx", "q_emb", "z_orig", "z_knns", "a_knns",
                                           "answer_aids", "a_emb_gt", "v_rank", "keep_mask")]


_PNAMES = ("answer_embedding", "w1", "b1", "w2", "b2", "w3", "b3", "w_out", "b_out")


class NcxParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _PNAMES]


class NcxGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _PNAMES]


class NcxMutanParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("wv", "bv", "wq", "bq", "whv", "bhv", "whq", "bhq", "wc", "bc")] + \
               [(n, C.c_int32) for n in ("dhv", "dhq", "R", "act_v", "act_q")]


class NcxError(RuntimeError):
    pass


_ERR = {-1: "NCX_E_NULL (required pointer is NULL)", -2: "NCX_E_DIMS (dimension out of range)",
        -3: "NCX_E_WORKSPACE (workspace too small or misaligned)", -4: "NCX_E_FLAGS (inconsistent lesion inputs)",
        -5: "NCX_E_UNSUPPORTED (RCCL could not be loaded)", -6: "NCX_E_COMM (RCCL reported an error)"}

_lib = None


def lib():
    """The loaded library; raises with a build hint when it is absent (never falls back)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NcxError("HIP library not built: %s is missing. Run `python -c 'import __graft_entry__ as g; "
                       "g.build()'` or `make -C vqa-counterexamples_amd`." % LIB_PATH)
    # The library shares the process's HIP runtime with PyTorch (streams and device pointers cross the boundary), and
    # PyTorch-ROCm ships its own libamdhip64: import torch FIRST so that the dynamic linker resolves this library's
    # libamdhip64.so.7 to the copy torch already loaded.  Loaded the other way round, the process ends up with two HIP
    # runtimes and every launch on a torch stream fails with hipErrorNoDevice.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    L.ncx_version.restype = C.c_char_p
    L.ncx_input_size.restype = C.c_int64
    L.ncx_input_size.argtypes = [C.POINTER(NcxDims)]
    L.ncx_workspace_bytes.restype = C.c_size_t
    L.ncx_workspace_bytes.argtypes = [C.POINTER(NcxDims)]
    L.ncx_forward.restype = C.c_int
    L.ncx_forward.argtypes = [C.POINTER(NcxDims), C.POINTER(NcxInputs), C.POINTER(NcxParams), C.c_void_p,
                              C.c_size_t, C.c_void_p, C.c_void_p]
    L.ncx_forward_phase.restype = C.c_int
    L.ncx_forward_phase.argtypes = [C.POINTER(NcxDims), C.POINTER(NcxInputs), C.POINTER(NcxParams), C.c_void_p,
                                    C.c_size_t, C.c_void_p, C.c_int32, C.c_void_p]
    L.ncx_loss_rank.restype = C.c_int
    L.ncx_loss_rank.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ncx_backward.restype = C.c_int
    L.ncx_backward.argtypes = [C.POINTER(NcxDims), C.POINTER(NcxInputs), C.POINTER(NcxParams), C.c_void_p,
                               C.c_size_t, C.c_void_p, C.POINTER(NcxGrads), C.c_void_p]
    L.ncx_backward_phase.restype = C.c_int
    L.ncx_backward_phase.argtypes = [C.POINTER(NcxDims), C.POINTER(NcxInputs), C.POINTER(NcxParams), C.c_void_p,
                                     C.c_size_t, C.c_void_p, C.POINTER(NcxGrads), C.c_int32, C.c_void_p]
    L.ncx_adam_step.restype = C.c_int
    L.ncx_adam_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float,
                                C.c_float, C.c_float, C.c_float, C.c_int32, C.c_float, C.c_void_p]
    L.ncx_vqa_workspace_bytes.restype = C.c_size_t
    L.ncx_vqa_workspace_bytes.argtypes = [C.POINTER(NcxDims), C.POINTER(NcxMutanParams)]
    L.ncx_vqa_forward.restype = C.c_int
    L.ncx_vqa_forward.argtypes = [C.POINTER(NcxDims), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(NcxMutanParams), C.c_void_p,
                                  C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ncx_knn_workspace_bytes.restype = C.c_size_t
    L.ncx_knn_workspace_bytes.argtypes = [C.c_int32, C.c_int32]
    L.ncx_knn.restype = C.c_int
    L.ncx_knn.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                          C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ncx_ws_region.restype = C.c_int
    L.ncx_ws_region.argtypes = [C.POINTER(NcxDims), C.c_int32, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.ncx_profile_begin.restype = C.c_int
    L.ncx_profile_begin.argtypes = [C.c_uint32, C.c_int32]
    L.ncx_profile_end.restype = C.c_int
    L.ncx_profile_end.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_int32]
    L.ncx_profile_stamps.restype = C.c_int
    L.ncx_profile_stamps.argtypes = [C.c_void_p, C.c_int64]
    L.ncx_train_tail.restype = C.c_int
    L.ncx_train_tail.argtypes = [C.POINTER(NcxDims), C.POINTER(NcxParams), C.c_void_p, C.c_size_t] + [C.c_void_p] * 7 + [C.POINTER(NcxGrads), C.c_void_p]
    L.ncx_comm_unique_id.restype = C.c_int; L.ncx_comm_unique_id.argtypes = [C.c_void_p]
    L.ncx_comm_create.restype = C.c_int; L.ncx_comm_create.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    L.ncx_comm_destroy.restype = C.c_int; L.ncx_comm_destroy.argtypes = [C.c_void_p]
    L.ncx_allreduce.restype = C.c_int; L.ncx_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.ncx_plan_query.restype = C.c_int
    L.ncx_plan_query.argtypes = [C.POINTER(NcxDims), C.c_int32, C.POINTER(C.c_int32)]
    _lib = L
    return L


def check(rc, what):
    if rc == 0:
        return
    if rc < 0:
        raise NcxError("%s failed: %s" % (what, _ERR.get(rc, rc)))
    raise NcxError("%s failed: hipError_t %d" % (what, rc))


def version():
    return lib().ncx_version().decode()


GEMM_IDS = dict(GT=0, SH=1, MAIN=2, FWD_L=3, DW1C=4, DW1S=5, DE=6, DW1AK=7, DAGT=8, DWL=9, DXL=10)


def profile_begin(names, max_launches=4096):
    mask = 0
    for n in names:
        mask |= 1 << GEMM_IDS[n]
    check(lib().ncx_profile_begin(mask, max_launches), "ncx_profile_begin")


def profile_end(cap=4096):
    """-> {gemm name: [ms, ...]} (synchronises the recorded events)."""
    ms = (C.c_float * cap)()
    ids = (C.c_int32 * cap)()
    n = lib().ncx_profile_end(ms, ids, cap)
    if n < 0:
        check(n, "ncx_profile_end")
    inv = {v: k for k, v in GEMM_IDS.items()}
    out = {}
    for i in range(n):
        out.setdefault(inv[ids[i]], []).append(float(ms[i]))
    return out


def profile_stamps(buf=None):
    """Arm (int64 device tensor, >= 16 words per workgroup of MAIN) or disarm (None) the in-kernel clock stamps of MAIN."""
    if buf is None:
        check(lib().ncx_profile_stamps(None, 0), "ncx_profile_stamps")
    else:
        check(lib().ncx_profile_stamps(C.c_void_p(buf.data_ptr()), buf.numel()), "ncx_profile_stamps")


def plan_query(d, name):
    out = (C.c_int32 * 6)()
    check(lib().ncx_plan_query(C.byref(d), GEMM_IDS[name], out), "ncx_plan_query")
    return dict(form=("NT", "TN", "NN")[out[0]], M=out[1], N=out[2], ksteps=out[3],
                tile=("64x64", "128x128", "96x128", "96x64", "128x64", "48x128", "48x64 (per-triplet fold of the two v segments)",
                      "96x64 (per-triplet fold of the two v segments, four triplets per workgroup)",
                      "192x64 (per-triplet fold of the two v segments, eight triplets per 8-wave workgroup, one workgroup per CU)")[out[4]], ksplit=out[5])


# ---- the C ABI's RCCL handle (include/neuralcx.h: ncx_comm_*, ncx_allreduce) ---------------------------------------------
# The engine exchanges gradients through torch.distributed (backend "nccl" = RCCL); these wrappers are the same collective
# for a host that binds the C ABI without torch.distributed, and what the GPU test drives.
def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    check(lib().ncx_comm_unique_id(buf), "ncx_comm_unique_id")
    return buf.raw


def comm_create(unique_id: bytes, nranks: int, rank: int) -> C.c_void_p:
    """Binds the CURRENT device (torch.cuda.set_device first).  Collective: every rank calls it with the same id."""
    if len(unique_id) != 128:
        raise ValueError("unique id must be the 128 bytes ncx_comm_unique_id produced on rank 0")
    comm = C.c_void_p()
    check(lib().ncx_comm_create(C.create_string_buffer(unique_id, 128), int(nranks), int(rank), C.byref(comm)), "ncx_comm_create")
    return comm


def allreduce(comm: C.c_void_p, t, stream_ptr=None) -> None:
    """In-place fp32 sum over the ranks of `comm`, ordered on the given HIP stream (default: torch's current stream)."""
    import torch
    if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
        raise TypeError("ncx_allreduce takes a contiguous fp32 device tensor")
    sp = torch.cuda.current_stream(t.device).cuda_stream if stream_ptr is None else stream_ptr
    check(lib().ncx_allreduce(comm, C.c_void_p(t.data_ptr()), t.numel(), C.c_void_p(sp)), "ncx_allreduce")


def comm_destroy(comm: C.c_void_p) -> None:
    check(lib().ncx_comm_destroy(comm), "ncx_comm_destroy")

