// src/ksched_sys.rs -- mechanical binding of include/ksched.h (ABI version 4): one `extern "C"` item per symbol the
// header declares, same order.  Nothing here allocates or panics.  tests/test_abi_symbols.py (in the ksched repository)
// checks this list against the header and the shared library's exports.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct ksched_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct ksched_pipe {
    _private: [u8; 0],
}
#[repr(C)]
pub struct ksched_comm {
    _private: [u8; 0],
}

pub const KSCHED_ABI_VERSION: u32 = 6;
pub const KSCHED_MAX_KEYS: u32 = 32;
pub const KSCHED_MAX_ATTEMPTS: u32 = 64;
pub const KSCHED_SEL_NEVER: u32 = 0xFFFF_FFFF;
pub const KSCHED_COMM_ID_BYTES: u32 = 128;

pub const KSCHED_OK: c_int = 0;
pub const KSCHED_E_INVAL: c_int = -1;
pub const KSCHED_E_NODEVICE: c_int = -2;
pub const KSCHED_E_HIP: c_int = -3;
pub const KSCHED_E_NOMEM: c_int = -4;
pub const KSCHED_E_STATE: c_int = -5;
pub const KSCHED_E_UNSUPPORTED: c_int = -6;
pub const KSCHED_E_RCCL: c_int = -7;

pub const KSCHED_FIT: u32 = 0x01;
pub const KSCHED_SEL: u32 = 0x02;
pub const KSCHED_TAINT: u32 = 0x04;
pub const KSCHED_PICK_SAMPLED: u32 = 0x08;
pub const KSCHED_PICK_BESTFIT: u32 = 0x10;
pub const KSCHED_WANT_FIT_MASK: u32 = 0x20;

pub const KSCHED_REASON_OK: c_int = 0;
pub const KSCHED_REASON_NOT_ENOUGH_RESOURCES: c_int = 1; // InvalidNodeReason::NotEnoughResources   (src/predicates.rs:16)
pub const KSCHED_REASON_NODE_SELECTOR_MISMATCH: c_int = 2; // InvalidNodeReason::NodeSelectorMismatch (src/predicates.rs:17)
pub const KSCHED_REASON_TAINT_NOT_TOLERATED: c_int = 3; // extension E2 only

pub const KSCHED_OPT_KERNEL: c_int = 1;
pub const KSCHED_OPT_TIMING: c_int = 2;
pub const KSCHED_OPT_DEBUG: c_int = 3;
pub const KSCHED_OPT_TRACE: c_int = 4;
pub const KSCHED_OPT_PICK_FROM_MASK: c_int = 5;
pub const KSCHED_OPT_INDEX_BUILD: c_int = 6;
pub const KSCHED_OPT_BESTFIT_STAGES: c_int = 7;
pub const KSCHED_OPT_SNAPSHOT_STREAM: c_int = 8;
pub const KSCHED_OPT_FUSED_PICK: c_int = 9;
pub const KSCHED_OPT_FAULT: c_int = 10;
pub const KSCHED_OPT_PIPE_MODE: c_int = 11;
pub const KSCHED_PIPE_MAX_STREAMS: u32 = 8;
pub const KSCHED_OPT_GRID_CUS: c_int = 12;
pub const KSCHED_MASK_ALLOC_AUTO: u32 = 0;
pub const KSCHED_MASK_ALLOC_PLAIN: u32 = 1;
pub const KSCHED_MASK_ALLOC_VMM: u32 = 2;
pub const KSCHED_MASK_ALLOC_PROBE: u32 = 11;
pub const KSCHED_OPT_MASK_PROBE: c_int = 13;
pub const KSCHED_OPT_ROUND_ORDER: c_int = 14;

extern "C" {
    // ---- lifetime
    pub fn ksched_create(out: *mut *mut ksched_ctx, device_id: c_int) -> c_int;
    pub fn ksched_destroy(ctx: *mut ksched_ctx);
    pub fn ksched_abi_version() -> u32;
    pub fn ksched_device_count() -> c_int;
    pub fn ksched_strerror(code: c_int) -> *const c_char;
    pub fn ksched_last_error(ctx: *const ksched_ctx) -> *const c_char;
    pub fn ksched_mask_words(n_nodes: u32) -> u32;
    pub fn ksched_set_option(ctx: *mut ksched_ctx, option: c_int, value: i64) -> c_int;
    // ---- node snapshot
    pub fn ksched_set_nodes(
        ctx: *mut ksched_ctx, n: u32, avail_cpu_milli: *const i64, avail_mem_bytes: *const i64, label_val_ids: *const u32,
        n_keys: u32, taints: *const u64,
    ) -> c_int;
    pub fn ksched_update_nodes(
        ctx: *mut ksched_ctx, count: u32, node_index: *const u32, avail_cpu_milli: *const i64, avail_mem_bytes: *const i64,
    ) -> c_int;
    pub fn ksched_forget_stream(ctx: *mut ksched_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn ksched_num_nodes(ctx: *const ksched_ctx) -> u32;
    pub fn ksched_num_keys(ctx: *const ksched_ctx) -> u32;
    // ---- evaluation
    pub fn ksched_eval(
        ctx: *mut ksched_ctx, p: u32, req_cpu_milli: *const i64, req_mem_bytes: *const i64, sel_val_ids: *const u32,
        tolerations: *const u64, samples: *const u32, attempts: u32, flags: u32, out_feasible: *mut u64, out_fit: *mut u64,
        out_binding: *mut i32,
    ) -> c_int;
    // ---- one host thread, several devices: ksched_eval in two halves
    pub fn ksched_shard_bounds(p: u32, nranks: u32, rank: u32, lo: *mut u32, hi: *mut u32, count_per_rank: *mut u32);
    pub fn ksched_eval_begin(
        ctx: *mut ksched_ctx, p: u32, req_cpu_milli: *const i64, req_mem_bytes: *const i64, sel_val_ids: *const u32, sel_stride: u32,
        tolerations: *const u64, samples: *const u32, attempts: u32, flags: u32, out_feasible: *mut u64, out_fit: *mut u64,
        binding_capacity: u32, binding_dev: *mut *mut i32, hip_stream: *mut *mut c_void,
    ) -> c_int;
    pub fn ksched_gather_buffer(ctx: *mut ksched_ctx, count: u32, dev: *mut *mut i32) -> c_int;
    pub fn ksched_eval_end(ctx: *mut ksched_ctx, bindings_dev: *const i32, count: u32, out_host: *mut i32) -> c_int;
    pub fn ksched_eval_device(
        ctx: *mut ksched_ctx, p: u32, req_cpu_milli: *const i64, req_mem_bytes: *const i64, sel_val_ids: *const u32,
        tolerations: *const u64, samples: *const u32, attempts: u32, flags: u32, out_feasible: *mut u64, out_fit: *mut u64,
        out_binding: *mut i32, hip_stream: *mut c_void,
    ) -> c_int;
    pub fn ksched_eval_device_pitched(
        ctx: *mut ksched_ctx, p: u32, req_cpu_milli: *const i64, req_mem_bytes: *const i64, sel_val_ids: *const u32,
        tolerations: *const u64, samples: *const u32, attempts: u32, flags: u32, out_feasible: *mut u64, out_fit: *mut u64,
        out_binding: *mut i32, mask_pitch_words: u32, hip_stream: *mut c_void,
    ) -> c_int;
    pub fn ksched_mask_pitch(n_nodes: u32) -> u32;
    pub fn ksched_mask_alloc(ctx: *mut ksched_ctx, p: u32, how: u32, out_mask: *mut *mut u64, out_pitch_words: *mut u32) -> c_int;
    pub fn ksched_mask_free(ctx: *mut ksched_ctx, mask: *mut u64) -> c_int;
    pub fn ksched_mask_probe_report(ctx: *mut ksched_ctx, out_us: *mut f64, cap: u32) -> c_int;
    pub fn ksched_pick_device(
        ctx: *mut ksched_ctx, p: u32, feasible: *const u64, mask_pitch_words: u32, req_mem_bytes: *const i64,
        samples: *const u32, attempts: u32, flags: u32, out_binding: *mut i32, hip_stream: *mut c_void,
    ) -> c_int;
    pub fn ksched_pick(
        ctx: *mut ksched_ctx, p: u32, feasible: *const u64, req_mem_bytes: *const i64, samples: *const u32, attempts: u32, flags: u32,
        out_binding: *mut i32,
    ) -> c_int;
    // ---- pipelined evaluation
    pub fn ksched_pipe_create(ctx: *mut ksched_ctx, depth: u32, out: *mut *mut ksched_pipe) -> c_int;
    pub fn ksched_pipe_destroy(pipe: *mut ksched_pipe);
    pub fn ksched_pipe_submit(
        pipe: *mut ksched_pipe, slot: u32, p: u32, req_cpu_milli: *const i64, req_mem_bytes: *const i64,
        sel_val_ids: *const u32, tolerations: *const u64, samples: *const u32, attempts: u32, flags: u32, mask: *mut u64,
        mask_pitch_words: u32, binding: *mut i32,
    ) -> c_int;
    pub fn ksched_pipe_wait(pipe: *mut ksched_pipe, slot: u32, hip_stream: *mut c_void) -> c_int;
    pub fn ksched_pipe_wait_mask(pipe: *mut ksched_pipe, slot: u32, hip_stream: *mut c_void) -> c_int;
    pub fn ksched_pipe_stream(pipe: *mut ksched_pipe, which: c_int) -> *mut c_void;
    pub fn ksched_pipe_slot_stream(pipe: *mut ksched_pipe, slot: u32) -> *mut c_void;
    // ---- reasons
    pub fn ksched_reason(feasible_row: *const u64, fit_row: *const u64, node: u32, flags: u32) -> c_int;
    pub fn ksched_explain(
        ctx: *mut ksched_ctx, p: u32, req_cpu_milli: *const i64, req_mem_bytes: *const i64, sel_val_ids: *const u32,
        tolerations: *const u64, count: u32, pair_pod: *const u32, pair_node: *const u32, flags: u32, out_reason: *mut i32,
    ) -> c_int;
    // ---- multi-GPU: RCCL all-gather of the (pod -> node) bindings
    pub fn ksched_comm_unique_id(id: *mut u8) -> c_int;
    pub fn ksched_comm_create(ctx: *mut ksched_ctx, id: *const u8, rank: c_int, nranks: c_int, out: *mut *mut ksched_comm) -> c_int;
    pub fn ksched_comm_create_local(ctxs: *const *mut ksched_ctx, n: c_int, out: *mut *mut ksched_comm) -> c_int;
    pub fn ksched_comm_destroy(comm: *mut ksched_comm);
    pub fn ksched_comm_rank(comm: *const ksched_comm) -> c_int;
    pub fn ksched_comm_size(comm: *const ksched_comm) -> c_int;
    pub fn ksched_allgather_bindings(
        comm: *mut ksched_comm, local: *const i32, gathered: *mut i32, count_per_rank: u32, hip_stream: *mut c_void,
    ) -> c_int;
    pub fn ksched_allgather_bindings_local(
        comms: *const *mut ksched_comm, n: c_int, local: *const *const i32, gathered: *const *mut i32, count_per_rank: u32,
        hip_streams: *const *mut c_void,
    ) -> c_int;
    pub fn ksched_comm_last_error() -> *const c_char;
    // ---- measurement / diagnostics
    pub fn ksched_kernel_time_ms(ctx: *mut ksched_ctx, total_ms: *mut f64, launches: *mut u64) -> c_int;
    pub fn ksched_kernel_time_samples(ctx: *mut ksched_ctx, out_ms: *mut f64, cap: u32) -> c_int;
    pub fn ksched_index_checksum(ctx: *mut ksched_ctx, out: *mut u64) -> c_int;
    pub fn ksched_trace_read(ctx: *mut ksched_ctx, out: *mut u64, max_blocks: u32) -> c_int;
    pub fn ksched_last_kernel(ctx: *const ksched_ctx) -> *const c_char;
    pub fn ksched_last_pick(ctx: *const ksched_ctx) -> *const c_char;
}

/// Every entry point of the binding with its address in the library that was linked: building the table makes the linker
/// resolve all of them (a stale libksched_hip.so fails at link time, not at the first call of a rarely used function).
pub fn symbol_table() -> Vec<(&'static str, usize)> {
    return vec![
        ("ksched_create", ksched_create as usize),
        ("ksched_destroy", ksched_destroy as usize),
        ("ksched_abi_version", ksched_abi_version as usize),
        ("ksched_device_count", ksched_device_count as usize),
        ("ksched_strerror", ksched_strerror as usize),
        ("ksched_last_error", ksched_last_error as usize),
        ("ksched_mask_words", ksched_mask_words as usize),
        ("ksched_set_option", ksched_set_option as usize),
        ("ksched_set_nodes", ksched_set_nodes as usize),
        ("ksched_update_nodes", ksched_update_nodes as usize),
        ("ksched_forget_stream", ksched_forget_stream as usize),
        ("ksched_num_nodes", ksched_num_nodes as usize),
        ("ksched_num_keys", ksched_num_keys as usize),
        ("ksched_eval", ksched_eval as usize),
        ("ksched_shard_bounds", ksched_shard_bounds as usize),
        ("ksched_eval_begin", ksched_eval_begin as usize),
        ("ksched_gather_buffer", ksched_gather_buffer as usize),
        ("ksched_eval_end", ksched_eval_end as usize),
        ("ksched_eval_device", ksched_eval_device as usize),
        ("ksched_eval_device_pitched", ksched_eval_device_pitched as usize),
        ("ksched_mask_pitch", ksched_mask_pitch as usize),
        ("ksched_mask_alloc", ksched_mask_alloc as usize),
        ("ksched_mask_free", ksched_mask_free as usize),
        ("ksched_mask_probe_report", ksched_mask_probe_report as usize),
        ("ksched_pick_device", ksched_pick_device as usize),
        ("ksched_pick", ksched_pick as usize),
        ("ksched_pipe_create", ksched_pipe_create as usize),
        ("ksched_pipe_destroy", ksched_pipe_destroy as usize),
        ("ksched_pipe_submit", ksched_pipe_submit as usize),
        ("ksched_pipe_wait", ksched_pipe_wait as usize),
        ("ksched_pipe_wait_mask", ksched_pipe_wait_mask as usize),
        ("ksched_pipe_stream", ksched_pipe_stream as usize),
        ("ksched_pipe_slot_stream", ksched_pipe_slot_stream as usize),
        ("ksched_reason", ksched_reason as usize),
        ("ksched_explain", ksched_explain as usize),
        ("ksched_comm_unique_id", ksched_comm_unique_id as usize),
        ("ksched_comm_create", ksched_comm_create as usize),
        ("ksched_comm_create_local", ksched_comm_create_local as usize),
        ("ksched_comm_destroy", ksched_comm_destroy as usize),
        ("ksched_comm_rank", ksched_comm_rank as usize),
        ("ksched_comm_size", ksched_comm_size as usize),
        ("ksched_allgather_bindings", ksched_allgather_bindings as usize),
        ("ksched_allgather_bindings_local", ksched_allgather_bindings_local as usize),
        ("ksched_comm_last_error", ksched_comm_last_error as usize),
        ("ksched_kernel_time_ms", ksched_kernel_time_ms as usize),
        ("ksched_kernel_time_samples", ksched_kernel_time_samples as usize),
        ("ksched_index_checksum", ksched_index_checksum as usize),
        ("ksched_trace_read", ksched_trace_read as usize),
        ("ksched_last_kernel", ksched_last_kernel as usize),
        ("ksched_last_pick", ksched_last_pick as usize),
    ];
}

/// Every constant of the binding (tests/test_rust_overlay.py in the ksched repository compares the values with include/ksched.h).
pub fn constant_table() -> Vec<(&'static str, i64)> {
    return vec![
        ("KSCHED_ABI_VERSION", KSCHED_ABI_VERSION as i64),
        ("KSCHED_MAX_KEYS", KSCHED_MAX_KEYS as i64),
        ("KSCHED_MAX_ATTEMPTS", KSCHED_MAX_ATTEMPTS as i64),
        ("KSCHED_SEL_NEVER", KSCHED_SEL_NEVER as i64),
        ("KSCHED_COMM_ID_BYTES", KSCHED_COMM_ID_BYTES as i64),
        ("KSCHED_OK", KSCHED_OK as i64),
        ("KSCHED_E_INVAL", KSCHED_E_INVAL as i64),
        ("KSCHED_E_NODEVICE", KSCHED_E_NODEVICE as i64),
        ("KSCHED_E_HIP", KSCHED_E_HIP as i64),
        ("KSCHED_E_NOMEM", KSCHED_E_NOMEM as i64),
        ("KSCHED_E_STATE", KSCHED_E_STATE as i64),
        ("KSCHED_E_UNSUPPORTED", KSCHED_E_UNSUPPORTED as i64),
        ("KSCHED_E_RCCL", KSCHED_E_RCCL as i64),
        ("KSCHED_FIT", KSCHED_FIT as i64),
        ("KSCHED_SEL", KSCHED_SEL as i64),
        ("KSCHED_TAINT", KSCHED_TAINT as i64),
        ("KSCHED_PICK_SAMPLED", KSCHED_PICK_SAMPLED as i64),
        ("KSCHED_PICK_BESTFIT", KSCHED_PICK_BESTFIT as i64),
        ("KSCHED_WANT_FIT_MASK", KSCHED_WANT_FIT_MASK as i64),
        ("KSCHED_REASON_OK", KSCHED_REASON_OK as i64),
        ("KSCHED_REASON_NOT_ENOUGH_RESOURCES", KSCHED_REASON_NOT_ENOUGH_RESOURCES as i64),
        ("KSCHED_REASON_NODE_SELECTOR_MISMATCH", KSCHED_REASON_NODE_SELECTOR_MISMATCH as i64),
        ("KSCHED_REASON_TAINT_NOT_TOLERATED", KSCHED_REASON_TAINT_NOT_TOLERATED as i64),
        ("KSCHED_OPT_KERNEL", KSCHED_OPT_KERNEL as i64),
        ("KSCHED_OPT_TIMING", KSCHED_OPT_TIMING as i64),
        ("KSCHED_OPT_DEBUG", KSCHED_OPT_DEBUG as i64),
        ("KSCHED_OPT_TRACE", KSCHED_OPT_TRACE as i64),
        ("KSCHED_OPT_PICK_FROM_MASK", KSCHED_OPT_PICK_FROM_MASK as i64),
        ("KSCHED_OPT_INDEX_BUILD", KSCHED_OPT_INDEX_BUILD as i64),
        ("KSCHED_OPT_BESTFIT_STAGES", KSCHED_OPT_BESTFIT_STAGES as i64),
        ("KSCHED_OPT_SNAPSHOT_STREAM", KSCHED_OPT_SNAPSHOT_STREAM as i64),
        ("KSCHED_OPT_FUSED_PICK", KSCHED_OPT_FUSED_PICK as i64),
        ("KSCHED_OPT_FAULT", KSCHED_OPT_FAULT as i64),
        ("KSCHED_OPT_PIPE_MODE", KSCHED_OPT_PIPE_MODE as i64),
        ("KSCHED_PIPE_MAX_STREAMS", KSCHED_PIPE_MAX_STREAMS as i64),
        ("KSCHED_OPT_GRID_CUS", KSCHED_OPT_GRID_CUS as i64),
        ("KSCHED_MASK_ALLOC_AUTO", KSCHED_MASK_ALLOC_AUTO as i64),
        ("KSCHED_MASK_ALLOC_PLAIN", KSCHED_MASK_ALLOC_PLAIN as i64),
        ("KSCHED_MASK_ALLOC_VMM", KSCHED_MASK_ALLOC_VMM as i64),
        ("KSCHED_MASK_ALLOC_PROBE", KSCHED_MASK_ALLOC_PROBE as i64),
        ("KSCHED_OPT_MASK_PROBE", KSCHED_OPT_MASK_PROBE as i64),
        ("KSCHED_OPT_ROUND_ORDER", KSCHED_OPT_ROUND_ORDER as i64),
    ];
}
