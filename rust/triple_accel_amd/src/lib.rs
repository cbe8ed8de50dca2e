//! `triple_accel` public API (v0.4.0 surface: crate root re-exports + the `hamming` and `levenshtein` modules)
//! backed by the MI355X engine through its C ABI (`include/triple_accel_amd.h`).
//!
//! Callers keep `use triple_accel::*;` unchanged.  Differences, all documented in INTEGRATION.md:
//! * search functions return an eager iterator (the GPU scans the whole haystack at once);
//! * there is no CPU fallback below this layer: without a usable GPU the calls panic with the HIP status;
//! * tracebacks needing more than 8 GB of device records (about 140K x 140K bytes) are refused (`levenshtein_*_with_opts`
//!   panics there with the library's status);
//! * every function implements the reference's SCALAR rules.  On an AVX2/SSE4.1 host the reference's own
//!   `levenshtein_simd_k_with_opts` / `levenshtein_search_simd_with_opts` take its SIMD cores instead, whose transposition
//!   is an unconditional blend (src/levenshtein.rs:2384-2388) where the scalar path tests `<=` (:517-525), and whose
//!   tracebacks / match starts can differ from the scalar ones on ties.  Where the two disagree this crate returns the
//!   scalar answer under both names (INTEGRATION.md section 5 lists the cases);
//! * the generic `T: PartialEq` entry points map their items onto bytes: more than 254 distinct items COMMON to both strings panic.
use std::os::raw::{c_int, c_void};

#[derive(Debug, PartialEq)]
pub struct Match { pub start: usize, pub end: usize, pub k: u32 }
#[derive(Debug, PartialEq, Copy, Clone)]
pub enum EditType { Match, Mismatch, AGap, BGap, Transpose }
#[derive(Debug, PartialEq)]
pub struct Edit { pub edit: EditType, pub count: usize }
#[derive(Debug, PartialEq, Copy, Clone)]
pub enum SearchType { All, Best }

mod ffi {
    use super::*;
    #[repr(C)] #[derive(Copy, Clone)]
    pub struct TaEditCosts { pub mismatch_cost: u8, pub gap_cost: u8, pub start_gap_cost: u8, pub has_transpose: u8, pub transpose_cost: u8 }
    #[repr(C)] pub struct TaMatch { pub start: u64, pub end: u64, pub k: u32, pub pad_: u32 }
    #[repr(C)] pub struct TaEdit { pub edit: u32, pub pad_: u32, pub count: u64 }
    /// `ta_strings` (include/triple_accel_amd.h): string i of a batch is `blob[off[i] .. off[i+1])` (CSR, n + 1 device offsets) or,
    /// with `off` null, `blob[i * stride .. i * stride + len)`.  All pointers are DEVICE memory.
    #[repr(C)] #[derive(Copy, Clone)]
    pub struct TaStrings { pub blob: *const u8, pub off: *const u64, pub stride: u64, pub len: u64, pub max_len: u64 }
    pub const TA_NONE: u32 = 0xFFFF_FFFF;
    #[link(name = "triple_accel_amd")]
    extern "C" {
        pub fn ta_edit_costs_new(mismatch: u8, gap: u8, start_gap: u8, has_transpose: c_int, transpose: u8, out: *mut TaEditCosts) -> c_int;
        pub fn ta_edit_costs_check_search(c: *const TaEditCosts) -> c_int;
        pub fn ta_hamming(a: *const u8, a_len: usize, b: *const u8, b_len: usize, out: *mut u32) -> c_int;
        pub fn ta_levenshtein_simd_k_with_opts(a: *const u8, a_len: usize, b: *const u8, b_len: usize, k: u32, trace_on: c_int,
                                               costs: *const TaEditCosts, out: *mut u32) -> c_int;
        pub fn ta_levenshtein_trace(a: *const u8, a_len: usize, b: *const u8, b_len: usize, k: u32, costs: *const TaEditCosts,
                                    out: *mut u32, edits: *mut *mut TaEdit, n_edits: *mut usize) -> c_int;
        pub fn ta_levenshtein_exp_with_opts(a: *const u8, a_len: usize, b: *const u8, b_len: usize, trace_on: c_int,
                                            costs: *const TaEditCosts, out: *mut u32) -> c_int;
        pub fn ta_levenshtein_exp_trace(a: *const u8, a_len: usize, b: *const u8, b_len: usize, costs: *const TaEditCosts,
                                        out: *mut u32, edits: *mut *mut TaEdit, n_edits: *mut usize) -> c_int;
        pub fn ta_levenshtein_search_simd_with_opts(needle: *const u8, n: usize, haystack: *const u8, h: usize, k: u32,
                                                    search_type: c_int, costs: *const TaEditCosts, anchored: c_int,
                                                    out: *mut *mut TaMatch, n_out: *mut usize) -> c_int;
        pub fn ta_levenshtein_search_first(needle: *const u8, n: usize, haystack: *const u8, h: usize, k: u32,
                                           costs: *const TaEditCosts, anchored: c_int, out: *mut TaMatch, found: *mut c_int) -> c_int;
        pub fn ta_levenshtein_search_resume(needle: *const u8, n: usize, haystack: *const u8, h: usize, k: u32,
                                            costs: *const TaEditCosts, anchored: c_int, out: *mut *mut TaMatch, n_out: *mut usize) -> c_int;
        pub fn ta_hamming_search_simd_with_opts(needle: *const u8, n: usize, haystack: *const u8, h: usize, k: u32,
                                                search_type: c_int, out: *mut *mut TaMatch, n_out: *mut usize) -> c_int;
        pub fn ta_hamming_search_naive_with_opts(needle: *const u8, n: usize, haystack: *const u8, h: usize, k: u32,
                                                 search_type: c_int, out: *mut *mut TaMatch, n_out: *mut usize) -> c_int;
        pub fn ta_free(p: *mut c_void);
        pub fn ta_levenshtein_k_batch(a: *const TaStrings, b: *const TaStrings, n: usize, k: u32, costs: *const TaEditCosts,
                                      out_dev: *mut u32, stream: *mut c_void) -> c_int;
        pub fn ta_levenshtein_exp_batch(a: *const TaStrings, b: *const TaStrings, n: usize, costs: *const TaEditCosts,
                                        out_dev: *mut u32, stream: *mut c_void) -> c_int;
        pub fn ta_hamming_batch(a: *const TaStrings, b: *const TaStrings, n: usize, out_dev: *mut u32, stream: *mut c_void) -> c_int;
        pub fn ta_levenshtein_trace_batch(a: *const TaStrings, b: *const TaStrings, n: usize, k: u32, costs: *const TaEditCosts,
                                          out_dev: *mut u32, edits_dev: *mut TaEdit, n_edits_dev: *mut u32, cap: usize, stream: *mut c_void) -> c_int;
        pub fn ta_set_option(option: c_int, value: c_int) -> c_int;
        pub fn ta_thread_release();
        pub fn ta_device_count() -> c_int;
        pub fn ta_queue_create(k: u32, costs: *const TaEditCosts, out: *mut *mut c_void) -> c_int;
        pub fn ta_queue_push(q: *mut c_void, a: *const u8, a_len: usize, b: *const u8, b_len: usize, ticket: *mut usize) -> c_int;
        pub fn ta_queue_flush(q: *mut c_void, results: *mut *const u32, n: *mut usize) -> c_int;
        pub fn ta_queue_destroy(q: *mut c_void);
        pub fn ta_set_devices(devices: *const c_int, n: usize) -> c_int;
        pub fn ta_get_devices(out: *mut c_int, cap: usize, n_out: *mut usize) -> c_int;
        pub fn ta_levenshtein_k_batch_host(a_host: *const TaStrings, b_host: *const TaStrings, n: usize, k: u32, costs: *const TaEditCosts,
                                           out_host: *mut u32) -> c_int;
        pub fn ta_levenshtein_exp_batch_host(a_host: *const TaStrings, b_host: *const TaStrings, n: usize, costs: *const TaEditCosts,
                                             out_host: *mut u32) -> c_int;
        pub fn ta_hamming_batch_host(a_host: *const TaStrings, b_host: *const TaStrings, n: usize, out_host: *mut u32) -> c_int;
        pub fn ta_levenshtein_trace_batch_host(a_host: *const TaStrings, b_host: *const TaStrings, n: usize, k: u32, costs: *const TaEditCosts,
                                               out_host: *mut u32, packed_host: *mut u32, n_edits_host: *mut u32, cap: usize) -> c_int;
    }
    /// Frees what a thread holds inside the library (its stream, pinned buffers, device scratch) when the thread ends: the
    /// library itself frees nothing from a thread-exit hook.  Touched by every call through `check`.
    pub struct ThreadGuard;
    /// `ta_thread_release` blocks on all device work of the process (hipDeviceSynchronize) and frees the thread's stream, pinned buffer and
    /// scratch.  It is skipped when no device is reachable any more (process teardown: the HIP runtime may already be gone).
    impl Drop for ThreadGuard { fn drop(&mut self) { unsafe { if ta_device_count() > 0 { ta_thread_release() } } } }
    thread_local! { pub static THREAD_GUARD: ThreadGuard = ThreadGuard; }
    /// status codes -> the reference's panics (src/hamming.rs:318, src/lib.rs:240, src/levenshtein.rs:44-52,69)
    pub fn check(rc: c_int) {
        let _ = THREAD_GUARD.try_with(|_| ());      // (try_with: a call from another thread-local's destructor must not panic)
        match rc {
            0 => (),
            1 => panic!("assertion failed: a.len() == b.len()"),
            2 => panic!("No zero/null bytes allowed in the string!"),
            3 => panic!("invalid EditCosts"),
            8 => panic!("attempt to divide by zero"),
            _ => panic!("triple_accel_amd: status {} (no CPU fallback below the C ABI)", rc),
        }
    }
    pub unsafe fn take_matches(p: *mut TaMatch, n: usize) -> Vec<Match> {
        let v = (0..n).map(|i| { let m = &*p.add(i); Match { start: m.start as usize, end: m.end as usize, k: m.k } }).collect();
        ta_free(p as *mut c_void);
        v
    }
    pub unsafe fn take_edits(p: *mut TaEdit, n: usize) -> Vec<Edit> {
        const T: [EditType; 5] = [EditType::Match, EditType::Mismatch, EditType::AGap, EditType::BGap, EditType::Transpose];
        let v = (0..n).map(|i| { let e = &*p.add(i); Edit { edit: T[e.edit as usize], count: e.count as usize } }).collect();
        ta_free(p as *mut c_void);
        v
    }
}

pub mod hamming {
    use super::ffi::*;
    use super::*;

    /// src/hamming.rs:390
    pub fn hamming(a: &[u8], b: &[u8]) -> u32 {
        let mut out = 0u32;
        check(unsafe { ta_hamming(a.as_ptr(), a.len(), b.as_ptr(), b.len(), &mut out) });
        out
    }
    /// same result contract as `hamming` (src/hamming.rs:317, :354): one GPU kernel serves them all
    pub fn hamming_simd_parallel(a: &[u8], b: &[u8]) -> u32 { hamming(a, b) }
    pub fn hamming_simd_movemask(a: &[u8], b: &[u8]) -> u32 { hamming(a, b) }
    /// src/hamming.rs:36 (same assert, same count)
    pub fn hamming_naive(a: &[u8], b: &[u8]) -> u32 { hamming(a, b) }
    /// src/hamming.rs:176, :249 -- the word-wise CPU routines need `alloc_str` buffers there; here any slice will do
    pub fn hamming_words_64(a: &[u8], b: &[u8]) -> u32 { hamming(a, b) }
    pub fn hamming_words_128(a: &[u8], b: &[u8]) -> u32 { hamming(a, b) }

    /// src/hamming.rs:96 -- the scalar routine's contract: no NUL-byte panic, an empty needle matches at every offset
    pub fn hamming_search_naive_with_opts<'a>(needle: &'a [u8], haystack: &'a [u8], k: u32, search_type: SearchType)
        -> Box<dyn Iterator<Item = Match> + 'a> {
        let (mut p, mut n) = (std::ptr::null_mut::<TaMatch>(), 0usize);
        check(unsafe { ta_hamming_search_naive_with_opts(needle.as_ptr(), needle.len(), haystack.as_ptr(), haystack.len(), k,
                                                         (search_type == SearchType::Best) as c_int, &mut p, &mut n) });
        Box::new(unsafe { take_matches(p, n) }.into_iter())
    }
    /// src/hamming.rs:70
    pub fn hamming_search_naive<'a>(needle: &'a [u8], haystack: &'a [u8]) -> Box<dyn Iterator<Item = Match> + 'a> {
        hamming_search_naive_with_opts(needle, haystack, ((needle.len() as u32) >> 1) + ((needle.len() as u32) & 1), SearchType::Best)
    }

    /// src/hamming.rs:454
    pub fn hamming_search_simd_with_opts<'a>(needle: &'a [u8], haystack: &'a [u8], k: u32, search_type: SearchType)
        -> Box<dyn Iterator<Item = Match> + 'a> {
        let (mut p, mut n) = (std::ptr::null_mut::<TaMatch>(), 0usize);
        check(unsafe { ta_hamming_search_simd_with_opts(needle.as_ptr(), needle.len(), haystack.as_ptr(), haystack.len(), k,
                                                        (search_type == SearchType::Best) as c_int, &mut p, &mut n) });
        Box::new(unsafe { take_matches(p, n) }.into_iter())
    }
    /// src/hamming.rs:422
    pub fn hamming_search_simd<'a>(needle: &'a [u8], haystack: &'a [u8]) -> Box<dyn Iterator<Item = Match> + 'a> {
        hamming_search_simd_with_opts(needle, haystack, ((needle.len() as u32) >> 1) + ((needle.len() as u32) & 1), SearchType::Best)
    }
    /// src/hamming.rs:588
    pub fn hamming_search<'a>(needle: &'a [u8], haystack: &'a [u8]) -> Box<dyn Iterator<Item = Match> + 'a> {
        hamming_search_simd(needle, haystack)
    }
}

pub mod levenshtein {
    use super::ffi::*;
    use super::*;

    /// src/levenshtein.rs:20-26
    #[derive(Copy, Clone, Debug)]
    pub struct EditCosts { mismatch_cost: u8, gap_cost: u8, start_gap_cost: u8, transpose_cost: Option<u8> }

    impl EditCosts {
        /// src/levenshtein.rs:38-60 (the asserts run inside the library: TA_ERR_BAD_COSTS -> panic)
        pub fn new(mismatch_cost: u8, gap_cost: u8, start_gap_cost: u8, transpose_cost: Option<u8>) -> Self {
            let mut raw = TaEditCosts { mismatch_cost: 0, gap_cost: 0, start_gap_cost: 0, has_transpose: 0, transpose_cost: 0 };
            check(unsafe { ta_edit_costs_new(mismatch_cost, gap_cost, start_gap_cost, transpose_cost.is_some() as c_int,
                                             transpose_cost.unwrap_or(0), &mut raw) });
            Self { mismatch_cost, gap_cost, start_gap_cost, transpose_cost }
        }
        pub(crate) fn raw(&self) -> TaEditCosts {
            TaEditCosts { mismatch_cost: self.mismatch_cost, gap_cost: self.gap_cost, start_gap_cost: self.start_gap_cost,
                          has_transpose: self.transpose_cost.is_some() as u8, transpose_cost: self.transpose_cost.unwrap_or(0) }
        }
    }
    /// src/levenshtein.rs:76-89
    pub const LEVENSHTEIN_COSTS: EditCosts = EditCosts { mismatch_cost: 1, gap_cost: 1, start_gap_cost: 0, transpose_cost: None };
    pub const RDAMERAU_COSTS: EditCosts = EditCosts { mismatch_cost: 1, gap_cost: 1, start_gap_cost: 0, transpose_cost: Some(1) };

    /// src/levenshtein.rs:714
    pub fn levenshtein_simd_k_with_opts(a: &[u8], b: &[u8], k: u32, trace_on: bool, costs: EditCosts)
        -> Option<(u32, Option<Vec<Edit>>)> {
        let (mut out, c) = (0u32, costs.raw());
        if trace_on {
            let (mut p, mut n) = (std::ptr::null_mut::<TaEdit>(), 0usize);
            check(unsafe { ta_levenshtein_trace(a.as_ptr(), a.len(), b.as_ptr(), b.len(), k, &c, &mut out, &mut p, &mut n) });
            if out == TA_NONE { return None; }
            return Some((out, Some(unsafe { take_edits(p, n) })));
        }
        check(unsafe { ta_levenshtein_simd_k_with_opts(a.as_ptr(), a.len(), b.as_ptr(), b.len(), k, 0, &c, &mut out) });
        if out == TA_NONE { None } else { Some((out, None)) }
    }
    /// src/levenshtein.rs:677
    pub fn levenshtein_simd_k(a: &[u8], b: &[u8], k: u32) -> Option<u32> {
        levenshtein_simd_k_with_opts(a, b, k, false, LEVENSHTEIN_COSTS).map(|r| r.0)
    }
    /// src/levenshtein.rs:1397
    pub fn levenshtein(a: &[u8], b: &[u8]) -> u32 { levenshtein_simd_k(a, b, u32::MAX).unwrap() }
    /// src/levenshtein.rs:1419
    pub fn rdamerau(a: &[u8], b: &[u8]) -> u32 { levenshtein_simd_k_with_opts(a, b, u32::MAX, false, RDAMERAU_COSTS).unwrap().0 }
    /// src/levenshtein.rs:1480
    pub fn levenshtein_exp_with_opts(a: &[u8], b: &[u8], trace_on: bool, costs: EditCosts) -> (u32, Option<Vec<Edit>>) {
        let (mut out, c) = (0u32, costs.raw());
        if trace_on {
            let (mut p, mut n) = (std::ptr::null_mut::<TaEdit>(), 0usize);
            check(unsafe { ta_levenshtein_exp_trace(a.as_ptr(), a.len(), b.as_ptr(), b.len(), &c, &mut out, &mut p, &mut n) });
            return (out, Some(unsafe { take_edits(p, n) }));
        }
        check(unsafe { ta_levenshtein_exp_with_opts(a.as_ptr(), a.len(), b.as_ptr(), b.len(), 0, &c, &mut out) });
        (out, None)
    }
    /// src/levenshtein.rs:1445
    pub fn levenshtein_exp(a: &[u8], b: &[u8]) -> u32 { levenshtein_exp_with_opts(a, b, false, LEVENSHTEIN_COSTS).0 }
    /// src/levenshtein.rs:1516
    pub fn rdamerau_exp(a: &[u8], b: &[u8]) -> u32 { levenshtein_exp_with_opts(a, b, false, RDAMERAU_COSTS).0 }

    /// Pairs produced one at a time, answered together: a single call costs a kernel launch (22-25 us for a 256-byte pair against
    /// ~2 us on a host core); `push` copies a pair and returns its ticket, `flush` runs ONE batch pass over everything pushed and
    /// returns `levenshtein_simd_k_with_opts(a, b, k, false, costs)` of every pair in push order.  (No reference analogue.)
    pub struct Queue { q: *mut c_void }
    impl Queue {
        pub fn new(k: u32, costs: EditCosts) -> Self {
            let mut q = std::ptr::null_mut();
            check(unsafe { ta_queue_create(k, &costs.raw(), &mut q) });
            Queue { q }
        }
        pub fn push(&mut self, a: &[u8], b: &[u8]) -> usize {
            let mut t = 0usize;
            check(unsafe { ta_queue_push(self.q, a.as_ptr(), a.len(), b.as_ptr(), b.len(), &mut t) });
            t
        }
        pub fn flush(&mut self) -> Vec<Option<u32>> {
            let (mut p, mut n) = (std::ptr::null::<u32>(), 0usize);
            check(unsafe { ta_queue_flush(self.q, &mut p, &mut n) });
            (0..n).map(|i| { let v = unsafe { *p.add(i) }; if v == TA_NONE { None } else { Some(v) } }).collect()
        }
    }
    impl Drop for Queue { fn drop(&mut self) { unsafe { ta_queue_destroy(self.q) } } }

    /// What a caller's LOOP over `levenshtein_simd_k_with_opts(a, b, k, false, costs)` computes, answered by one batch pass per
    /// `flush_every` pairs: the drop-in single call costs a kernel launch (22-25 us for a 256-byte pair against ~2 us on a host core),
    /// a queued pair 1.0-1.8 us.  `Some(d)` / `None` per pair, in order.
    pub fn levenshtein_simd_k_with_opts_many<'a, I>(pairs: I, k: u32, costs: EditCosts) -> Vec<Option<u32>>
    where I: IntoIterator<Item = (&'a [u8], &'a [u8])> {
        const FLUSH_EVERY: usize = 1 << 16;
        let mut q = Queue::new(k, costs);
        let (mut out, mut pending) = (Vec::new(), 0usize);
        for (a, b) in pairs {
            q.push(a, b);
            pending += 1;
            if pending >= FLUSH_EVERY { out.extend(q.flush()); pending = 0; }
        }
        if pending > 0 { out.extend(q.flush()); }
        out
    }
    /// The same for pairs that already sit in slices: no queue copy -- the strings are gathered into one CSR blob per side and handed to
    /// the host-pointer batch entry, which shards them over the device set (`device::set_devices`; default: every visible GPU), each
    /// device uploading its contiguous slice through its own pinned ring and PCIe link (include/triple_accel_amd.h, "the device set").
    pub fn levenshtein_simd_k_with_opts_slices(pairs: &[(&[u8], &[u8])], k: u32, costs: EditCosts) -> Vec<Option<u32>> {
        let n = pairs.len();
        let (mut blob_a, mut blob_b) = (Vec::<u8>::new(), Vec::<u8>::new());
        let (mut off_a, mut off_b) = (Vec::<u64>::with_capacity(n + 1), Vec::<u64>::with_capacity(n + 1));
        off_a.push(0);
        off_b.push(0);
        for (a, b) in pairs {
            blob_a.extend_from_slice(a);
            blob_b.extend_from_slice(b);
            off_a.push(blob_a.len() as u64);
            off_b.push(blob_b.len() as u64);
        }
        let sa = TaStrings { blob: blob_a.as_ptr(), off: off_a.as_ptr(), stride: 0, len: 0, max_len: 0 };
        let sb = TaStrings { blob: blob_b.as_ptr(), off: off_b.as_ptr(), stride: 0, len: 0, max_len: 0 };
        let mut out = vec![0u32; n];
        check(unsafe { ta_levenshtein_k_batch_host(&sa, &sb, n, k, &costs.raw(), out.as_mut_ptr()) });
        out.into_iter().map(|v| if v == TA_NONE { None } else { Some(v) }).collect()
    }
    /// `pairs.map(|(a, b)| levenshtein(a, b))` through the queue
    pub fn levenshtein_many<'a, I>(pairs: I) -> Vec<u32>
    where I: IntoIterator<Item = (&'a [u8], &'a [u8])> {
        levenshtein_simd_k_with_opts_many(pairs, u32::MAX, LEVENSHTEIN_COSTS).into_iter().map(|d| d.expect("unbounded k")).collect()
    }

    /// All-mode result over a long haystack, lazily (the reference's iterator is lazy too, src/levenshtein.rs:2282-2420): the
    /// first element comes from `ta_levenshtein_search_first`, which stops scanning (and uploading) at the first window that
    /// holds a hit; the full search runs when a second element is asked for -- on the device copy the first call started, only the
    /// rest of the haystack is uploaded then (`ta_levenshtein_search_resume`).  Element for element the eager sequence.
    struct LazyAll<'a> { needle: &'a [u8], haystack: &'a [u8], k: u32, costs: EditCosts, anchored: bool, state: u8,
                         rest: std::vec::IntoIter<Match> }
    impl<'a> Iterator for LazyAll<'a> {
        type Item = Match;
        fn next(&mut self) -> Option<Match> {
            if self.state == 0 {
                self.state = 1;
                let (mut m, mut found, c) = (TaMatch { start: 0, end: 0, k: 0, pad_: 0 }, 0 as c_int, self.costs.raw());
                check(unsafe { ta_levenshtein_search_first(self.needle.as_ptr(), self.needle.len(), self.haystack.as_ptr(),
                                                           self.haystack.len(), self.k, &c, self.anchored as c_int, &mut m, &mut found) });
                if found == 0 { self.state = 2; return None; }
                return Some(Match { start: m.start as usize, end: m.end as usize, k: m.k });
            }
            if self.state == 1 {
                self.state = 2;
                let (mut p, mut n, c) = (std::ptr::null_mut::<TaMatch>(), 0usize, self.costs.raw());
                // (the haystack is borrowed for 'a: the same pointer, the same bytes as in the first call -- only what that call did not
                // upload travels now)
                check(unsafe { ta_levenshtein_search_resume(self.needle.as_ptr(), self.needle.len(), self.haystack.as_ptr(),
                                                            self.haystack.len(), self.k, &c, self.anchored as c_int, &mut p, &mut n) });
                self.rest = unsafe { take_matches(p, n) }.into_iter();
                self.rest.next();                              // the element already handed out
            }
            self.rest.next()
        }
    }
    const LAZY_SEARCH_FROM: usize = 1 << 20;

    /// src/levenshtein.rs:1911
    pub fn levenshtein_search_simd_with_opts<'a>(needle: &'a [u8], haystack: &'a [u8], k: u32, search_type: SearchType,
                                                  costs: EditCosts, anchored: bool) -> Box<dyn Iterator<Item = Match> + 'a> {
        if search_type == SearchType::All && haystack.len() >= LAZY_SEARCH_FROM && !needle.is_empty() {
            check(unsafe { ta_edit_costs_check_search(&costs.raw()) });    // the call's own panic stays eager (:1965)
            return Box::new(LazyAll { needle, haystack, k, costs, anchored, state: 0, rest: Vec::new().into_iter() });
        }
        let (mut p, mut n, c) = (std::ptr::null_mut::<TaMatch>(), 0usize, costs.raw());
        check(unsafe { ta_levenshtein_search_simd_with_opts(needle.as_ptr(), needle.len(), haystack.as_ptr(), haystack.len(), k,
                                                            (search_type == SearchType::Best) as c_int, &c, anchored as c_int,
                                                            &mut p, &mut n) });
        Box::new(unsafe { take_matches(p, n) }.into_iter())
    }
    /// src/levenshtein.rs:1866
    pub fn levenshtein_search_simd<'a>(needle: &'a [u8], haystack: &'a [u8]) -> Box<dyn Iterator<Item = Match> + 'a> {
        levenshtein_search_simd_with_opts(needle, haystack, ((needle.len() >> 1) as u32) + ((needle.len() as u32) & 1),
                                          SearchType::Best, LEVENSHTEIN_COSTS, false)
    }
    /// src/levenshtein.rs:2508
    pub fn levenshtein_search<'a>(needle: &'a [u8], haystack: &'a [u8]) -> Box<dyn Iterator<Item = Match> + 'a> {
        levenshtein_search_simd(needle, haystack)
    }

    // ---- the scalar entry points: the kernels implement the scalar rules, so these are the same calls under the other names

    /// items of any `T: PartialEq` -> bytes.  The recurrence only ever compares an item of `a` with an item of `b`, so the items
    /// that occur on ONE side only share one code per side (254: only in `a`, 255: only in `b`): any pair with at most 254
    /// distinct items COMMON to both strings rides the byte kernels.
    fn symbols<'x, T: PartialEq>(a: &'x [T], b: &'x [T]) -> (Vec<u8>, Vec<u8>) {
        let mut common: Vec<&'x T> = Vec::new();
        for x in a {
            if b.iter().any(|y| *y == *x) && !common.iter().any(|d| **d == *x) {
                assert!(common.len() < 254, "triple_accel_amd: more than 254 distinct symbols common to both strings cannot be mapped onto the byte kernels");
                common.push(x);
            }
        }
        let code = |x: &T, other: u8| common.iter().position(|d| **d == *x).map(|i| i as u8).unwrap_or(other);
        (a.iter().map(|x| code(x, 254)).collect(), b.iter().map(|x| code(x, 255)).collect())
    }
    /// src/levenshtein.rs:148
    pub fn levenshtein_naive_with_opts<T: PartialEq>(a: &[T], b: &[T], trace_on: bool, costs: EditCosts) -> (u32, Option<Vec<Edit>>) {
        let (ta, tb) = symbols(a, b);
        levenshtein_simd_k_with_opts(&ta, &tb, u32::MAX, trace_on, costs).unwrap()
    }
    /// src/levenshtein.rs:105
    pub fn levenshtein_naive<T: PartialEq>(a: &[T], b: &[T]) -> u32 { levenshtein_naive_with_opts(a, b, false, LEVENSHTEIN_COSTS).0 }
    /// src/levenshtein.rs:123 (the reference's spelling)
    pub fn levenstein_naive_str(a: &str, b: &str) -> u32 {
        let (a, b): (Vec<char>, Vec<char>) = (a.chars().collect(), b.chars().collect());
        levenshtein_naive(&a, &b)
    }
    /// src/levenshtein.rs:376
    pub fn levenshtein_naive_k_with_opts<T: PartialEq>(a: &[T], b: &[T], k: u32, trace_on: bool, costs: EditCosts)
        -> Option<(u32, Option<Vec<Edit>>)> {
        let (ta, tb) = symbols(a, b);
        levenshtein_simd_k_with_opts(&ta, &tb, k, trace_on, costs)
    }
    /// src/levenshtein.rs:342
    pub fn levenshtein_naive_k(a: &[u8], b: &[u8], k: u32) -> Option<u32> { levenshtein_simd_k(a, b, k) }
    /// src/levenshtein.rs:609-651: ASCII strings as they are, anything else through a table of at most 256 distinct chars
    pub fn levenshtein_simd_k_str(a: &str, b: &str, k: u32) -> Option<u32> {
        if a.is_ascii() && b.is_ascii() {
            return levenshtein_simd_k(a.as_bytes(), b.as_bytes(), k);
        }
        let mut chars: Vec<char> = Vec::with_capacity(256);
        let mut translate = |s: &str| -> Option<Vec<u8>> {
            s.chars().map(|c| match chars.iter().position(|&d| c == d) {
                Some(i) => Some(i as u8),
                None => { let idx = chars.len(); if idx < 256 { chars.push(c); Some(idx as u8) } else { None } }
            }).collect()
        };
        let ta = translate(a)?;
        let tb = translate(b)?;
        levenshtein_simd_k(&ta, &tb, k)
    }
    /// src/levenshtein.rs:1589
    pub fn levenshtein_search_naive_with_opts<'a>(needle: &'a [u8], haystack: &'a [u8], k: u32, search_type: SearchType,
                                                   costs: EditCosts, anchored: bool) -> Box<dyn Iterator<Item = Match> + 'a> {
        levenshtein_search_simd_with_opts(needle, haystack, k, search_type, costs, anchored)
    }
    /// src/levenshtein.rs:1549
    pub fn levenshtein_search_naive<'a>(needle: &'a [u8], haystack: &'a [u8]) -> Box<dyn Iterator<Item = Match> + 'a> {
        levenshtein_search_simd(needle, haystack)
    }
}

/// Batches of pairs RESIDENT IN HBM (no reference analogue: the reference answers one pair per call).  The caller owns the device
/// buffers (any HIP allocation: `hipMalloc`, a torch tensor's `data_ptr()`); `stream` is a `hipStream_t` (null = the default
/// stream); the calls enqueue kernels and return, results are device memory too (`u32` per pair, `0xFFFF_FFFF` = `None`).
pub mod device {
    use super::ffi::*;
    use super::levenshtein::EditCosts;
    use super::*;
    pub use super::ffi::{TaEdit, TaStrings};
    use std::os::raw::{c_int, c_void};

    /// Options of the calling thread (include/triple_accel_amd.h; off by default, never a change of an answer -- only of how much work a
    /// batch costs): `OPT_EARLY_OUT` -- wavefronts of fixed-length unit-cost batches stop once none of their pairs can end at or below k;
    /// `OPT_UNIT_PREFILTER` -- batches under weighted `EditCosts` run the unit-cost pass first and price only the pairs it could not rule out.
    pub const OPT_EARLY_OUT: c_int = 1;
    pub const OPT_UNIT_PREFILTER: c_int = 2;
    pub fn set_option(option: c_int, on: bool) { unsafe { check(ta_set_option(option, on as c_int)); } }

    /// The GPUs the host entry points fan out over (empty slice: every visible device, the default).  With more than one entry the
    /// `*_many` / `*_slices` batch functions, `Queue::flush` and the searches over haystacks of >= 8 MiB are partitioned over the set
    /// inside the library; callers of the reference's functions change nothing.  An id may be listed more than once.
    pub fn set_devices(devices: &[c_int]) { unsafe { check(ta_set_devices(devices.as_ptr(), devices.len())); } }
    pub fn get_devices() -> Vec<c_int> {
        let mut n = 0usize;
        unsafe { check(ta_get_devices(std::ptr::null_mut(), 0, &mut n)); }
        let mut v = vec![0 as c_int; n];
        unsafe { check(ta_get_devices(v.as_mut_ptr(), n, &mut n)); }
        v
    }

    /// N x `levenshtein_simd_k_with_opts(a_i, b_i, k, false, costs)`
    pub unsafe fn levenshtein_k_batch(a: &TaStrings, b: &TaStrings, n: usize, k: u32, costs: EditCosts, out_dev: *mut u32, stream: *mut c_void) {
        check(ta_levenshtein_k_batch(a, b, n, k, &costs.raw(), out_dev, stream));
    }
    /// N x `levenshtein_exp_with_opts(a_i, b_i, false, costs)`
    pub unsafe fn levenshtein_exp_batch(a: &TaStrings, b: &TaStrings, n: usize, costs: EditCosts, out_dev: *mut u32, stream: *mut c_void) {
        check(ta_levenshtein_exp_batch(a, b, n, &costs.raw(), out_dev, stream));
    }
    /// N x `hamming(a_i, b_i)` (`0xFFFF_FFFF` where the lengths differ)
    pub unsafe fn hamming_batch(a: &TaStrings, b: &TaStrings, n: usize, out_dev: *mut u32, stream: *mut c_void) {
        check(ta_hamming_batch(a, b, n, out_dev, stream));
    }
    /// N x `levenshtein_simd_k_with_opts(a_i, b_i, k, true, costs)`: distances, run counts and `cap` `TaEdit` records per pair
    pub unsafe fn levenshtein_trace_batch(a: &TaStrings, b: &TaStrings, n: usize, k: u32, costs: EditCosts, out_dev: *mut u32,
                                          edits_dev: *mut TaEdit, n_edits_dev: *mut u32, cap: usize, stream: *mut c_void) {
        check(ta_levenshtein_trace_batch(a, b, n, k, &costs.raw(), out_dev, edits_dev, n_edits_dev, cap, stream));
    }
}

/// src/lib.rs:197 (the reference pads and aligns for its u128 Hamming routines; the bytes are what matters here)
pub fn alloc_str(len: usize) -> Vec<u8> { vec![0u8; len] }
/// src/lib.rs:229
pub fn fill_str(dest: &mut [u8], src: &[u8]) {
    assert!(dest.len() >= src.len());
    dest[..src.len()].copy_from_slice(src);
}

// src/lib.rs:126-127
pub use hamming::{hamming, hamming_search};
pub use levenshtein::{levenshtein, levenshtein_exp, levenshtein_search, rdamerau, rdamerau_exp};
