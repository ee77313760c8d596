"""CPU-side checks of the C-ABI boundary: the library builds for gfx950, loads, and exports every symbol
include/hdyolo.h declares with a ctypes signature registered in hd_yolo_amd/_lib.py.  No compute calls."""
import os
import re

import pytest

from hd_yolo_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        build.build(verbose=False)
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'hdyolo.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(hdy_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_exported_and_bound(lib):
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in include/hdyolo.h but not exported'
        assert s in _lib.SIGNATURES, f'{s} has no ctypes signature'
    assert sorted(_lib.SIGNATURES) == syms


def test_pure_host_queries(lib):
    # one ABI revision in three places: the header's macro, the library built from it, the binding's SIGNATURES (load() refuses any other library)
    header = open(os.path.join(ROOT, 'include', 'hdyolo.h')).read()
    assert int(re.search(r'#define\s+HDY_ABI_VERSION\s+(\d+)', header).group(1)) == _lib.ABI_VERSION == lib.hdy_version()
    assert lib.hdy_conv_out_dim(640, 6, 2, 2) == 320 and lib.hdy_conv_out_dim(80, 3, 2, 1) == 40
    assert lib.hdy_conv_mtiles(129) == 2
    # statistic slabs: one per 128 output pixels (generic kernel) or one per workgroup (filter-resident 3x3 C=64 kernel)
    assert lib.hdy_conv_stat_slabs(2, 20, 20, 32, 32, 3, 3, 1, 1, _lib.BF16) == 7
    assert lib.hdy_conv_stat_slabs(64, 80, 80, 64, 64, 3, 3, 1, 1, _lib.BF16) == 512      # two workgroups per CU
    assert lib.hdy_conv_stat_slabs(64, 80, 80, 64, 64, 3, 3, 1, 1, _lib.F32) == 640      # generic kernel, one column tile: 768 resident workgroups x 5 tiles
    assert lib.hdy_conv_stat_slabs(64, 20, 20, 128, 256, 1, 1, 1, 0, _lib.BF16) == 100   # deep-pipelined kernel: one slab per workgroup position (100 row tiles of 256)
    with _lib.option('HDY_NO_DEEP', 1):
        assert lib.hdy_conv_stat_slabs(64, 20, 20, 128, 256, 1, 1, 1, 0, _lib.BF16) == 200   # generic kernel, two column tiles: one slab per 128 rows
    # packed sizes: rows padded to the N tile, K-extent padded to 128 bytes
    assert lib.hdy_conv_pack_elems(39, 128, 1, 1, 1, 0, _lib.PACK_FWD, _lib.BF16) == 64 * 128
    assert lib.hdy_conv_pack_elems(64, 64, 3, 3, 1, 1, _lib.PACK_FWD, _lib.F32) == 64 * 576
    # stride-2 dgrad = four parity classes holding 1+2+2+4 = 9 taps in total
    assert lib.hdy_conv_pack_elems(128, 64, 3, 3, 2, 1, _lib.PACK_DGRAD, _lib.BF16) == 64 * 128 * 9
    assert lib.hdy_nms_workspace_bytes(2, 25200) == 2 * 32768 * 8
    # up to 4096 kept boxes the list lives in LDS; beyond, 20 bytes per entry behind the sort keys (any max_det is served, utils_general.py:342)
    assert lib.hdy_nms_workspace_bytes_for(2, 25200, 300) == 2 * 32768 * 8 and lib.hdy_nms_workspace_bytes_for(2, 25200, 4096) == 2 * 32768 * 8
    assert lib.hdy_nms_workspace_bytes_for(2, 25200, 5000) == 2 * 32768 * 8 + 2 * 5000 * 20
    assert lib.hdy_bn_bwd_blocks(10) == 1 and lib.hdy_bn_bwd_blocks(10 ** 7) == 1024


def test_options_and_dispatch_log_are_host_state(lib):
    """kernel-selection switches: known names round-trip, unknown names fail with a status; the dispatch log starts empty on a thread"""
    assert lib.hdy_get_option(b'HDY_WGRAD_BLOCKS') == 512 and lib.hdy_get_option(b'HDY_TILE_INTERLEAVE') == 1
    prev = lib.hdy_set_option(b'HDY_NO_CONV3X3', 1)
    assert prev == 0 and lib.hdy_get_option(b'HDY_NO_CONV3X3') == 1
    # the switch steers the sizing query too: the filter-resident kernel's one-slab-per-workgroup count is gone
    assert lib.hdy_conv_stat_slabs(64, 80, 80, 64, 64, 3, 3, 1, 1, _lib.BF16) != 512
    assert lib.hdy_set_option(b'HDY_NO_CONV3X3', prev) == 1
    assert lib.hdy_conv_stat_slabs(64, 80, 80, 64, 64, 3, 3, 1, 1, _lib.BF16) == 512
    assert lib.hdy_set_option(b'HDY_NO_SUCH_THING', 1) < 0 and b'unknown option' in lib.hdy_last_error()
    with _lib.option('HDY_NO_BIG_TILES', 1):
        assert lib.hdy_get_option(b'HDY_NO_BIG_TILES') == 1
    assert lib.hdy_get_option(b'HDY_NO_BIG_TILES') == 0
    lib.hdy_dispatch_log_reset()
    assert lib.hdy_last_dispatch() == b'' and _lib.dispatch_log() == []


def test_invalid_arguments_return_status_not_crash(lib):
    rc = lib.hdy_conv_fwd(None, 8, None, None, None, None, 0, None, 8, None, 0, 1, 4, 4, 8, 8, 1, 1, 1, 0, 0, 0, _lib.BF16, 0, 0, None)
    assert rc < 0 and b'null' in lib.hdy_last_error()
    rc = lib.hdy_nms_batched(None, 1, 4, 7, 1, 0.1, 2.0, 10, 2.0, 0, None, None, None, None, None, None, None, None, 0, None)
    assert rc < 0


FAKE = 0x10000      # a 16-byte aligned non-NULL "device pointer": the calls below must fail validation before anything dereferences or launches


def _conv_fwd_with_stats(lib, slabs, N=64, H=20, W=20, C=128, K=256):
    return lib.hdy_conv_fwd(FAKE, C, FAKE, None, None, None, 0, FAKE, K, FAKE, slabs, N, H, W, C, K, 1, 1, 1, 0, 0, 0, _lib.BF16, 0, 0, None)


def test_statistic_slab_count_is_checked_against_the_kernel_about_to_run(lib):
    """VERDICT r03 'ABI hole': hdy_conv_fwd took `float* stats` without a size and decided the slab count at launch from the process-wide
    option table.  Now the caller passes the count it sized the array with; a switch flipped between hdy_conv_stat_slabs and the launch
    gives HDY_EINVAL (no launch, no out-of-bounds slab) — checked here on the host, in both directions."""
    shape = (64, 20, 20, 128, 256, 1, 1, 1, 0, _lib.BF16)
    deep = lib.hdy_conv_stat_slabs(*shape)
    with _lib.option('HDY_NO_DEEP', 1):
        generic = lib.hdy_conv_stat_slabs(*shape)
        assert (deep, generic) == (100, 200)
        # sized for the deep-pipelined kernel, launched after the switch: the generic kernel would write 200 slabs into room for 100
        rc = _conv_fwd_with_stats(lib, deep)
        assert rc == _lib.EINVAL and b'holds 100 slabs' in lib.hdy_last_error() and b'writes 200' in lib.hdy_last_error()
    # sized for the generic kernel, launched on the deep-pipelined one: 100 of the 200 slabs hdy_bn_finalize reads would be stale
    rc = _conv_fwd_with_stats(lib, generic)
    assert rc == _lib.EINVAL and b'holds 200 slabs' in lib.hdy_last_error() and b'writes 100' in lib.hdy_last_error()
    # the filter-resident 3x3 kernel (one slab per workgroup) against its switch
    s3 = (64, 80, 80, 64, 64, 3, 3, 1, 1, _lib.BF16)
    own = lib.hdy_conv_stat_slabs(*s3)
    with _lib.option('HDY_NO_CONV3X3', 1):
        rc = lib.hdy_conv_fwd(FAKE, 64, FAKE, None, None, None, 0, FAKE, 64, FAKE, own, 64, 80, 80, 64, 64, 3, 3, 1, 1, 0, 0, _lib.BF16, 0, 0, None)
        assert rc == _lib.EINVAL and b'slabs' in lib.hdy_last_error()
    assert lib.hdy_conv_fwd(FAKE, 64, FAKE, None, None, None, 0, FAKE, 64, FAKE, 0, 64, 80, 80, 64, 64, 3, 3, 1, 1, 0, 0, _lib.BF16, 0, 0, None) == _lib.EINVAL


def test_workspace_sizes_are_checked(lib):
    """every workspace pointer of the ABI travels with its size: a buffer smaller than the matching *_workspace_* query is HDY_EINVAL"""
    M, K = 64 * 20 * 20, 256
    need = lib.hdy_bn_bwd_workspace_bytes(M, K)
    assert need == (lib.hdy_bn_bwd_blocks(M) * 2 * K + 2 * K) * 4
    rc = lib.hdy_bn_act_bwd(FAKE, K, FAKE, K, FAKE, FAKE, FAKE, FAKE, FAKE, K, FAKE, FAKE, 0, M, K, 1, _lib.BF16, FAKE, need - 4, None)
    assert rc == _lib.EINVAL and b'workspace' in lib.hdy_last_error()
    rc = lib.hdy_bn_act_bwd_pair(FAKE, 128, FAKE, 128, 128, FAKE, K, FAKE, FAKE, FAKE, FAKE, FAKE, K, FAKE, FAKE, FAKE, FAKE, 0, M, K, 1, _lib.BF16, FAKE, need - 4, None)
    assert rc == _lib.EINVAL and b'workspace' in lib.hdy_last_error()
    assert lib.hdy_colsum(FAKE, 40, M, 40, FAKE, 0, _lib.BF16, FAKE, lib.hdy_colsum_workspace_bytes(M, 40) - 4, None) == _lib.EINVAL
    mt = 2000
    fin = lib.hdy_bn_finalize_workspace_bytes(mt, K)
    assert fin > 0
    rc = lib.hdy_bn_finalize(FAKE, K, mt, K, M, FAKE, FAKE, FAKE, FAKE, 1e-3, 0.03, FAKE, FAKE, FAKE, FAKE, FAKE, fin - 8, None)
    assert rc == _lib.EINVAL and b'workspace' in lib.hdy_last_error()
    gn = lib.hdy_groupnorm_workspace_floats(2, 64)
    assert lib.hdy_groupnorm_fwd(FAKE, 64, FAKE, FAKE, FAKE, 64, FAKE, FAKE, 2, 100, 64, 32, 1e-5, 1, _lib.BF16, FAKE, gn - 1, None) == _lib.EINVAL
    sd = lib.hdy_softdice_workspace_floats(2, 3)
    assert lib.hdy_softdice(FAKE, 4, FAKE, None, 2, 100, 3, FAKE, None, None, 4, FAKE, sd - 1, None) == _lib.EINVAL
    assert lib.hdy_softdice_wgrad(FAKE, FAKE, None, 2, 10, 10, 3, 5, FAKE, FAKE, FAKE, sd - 1, None) == _lib.EINVAL


def test_fastdiv_reciprocal_is_exact(lib):
    """conv_igemm.hip recovers (image, row, column) from a flattened output index with n / d == mulhi(2n, magic) >> shift; the identity
    must hold for every n < 2^31 the kernels can see: checked at the multiples of d and their neighbours, the top of the range and at
    random points, for the divisors that occur (image sides, pixel counts, channel counts, tap-window widths)."""
    import ctypes
    import random
    rng = random.Random(0)
    ds = list(range(1, 70)) + [80, 96, 128, 160, 192, 320, 640, 1024, 1280, 20 * 20, 40 * 40, 80 * 80, 160 * 160, 320 * 320, 512 * 512, 1024 * 1024,
                                1023, 1025, 65535, 65537, 999983] + [rng.randrange(2, 1 << 22) for _ in range(200)]
    for d in ds:
        mg, sh = ctypes.c_uint(), ctypes.c_int()
        assert lib.hdy_fastdiv_magic(d, ctypes.byref(mg), ctypes.byref(sh)) == 0
        pts = [0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, (1 << 31) - 1, ((1 << 31) - 1) // d * d, ((1 << 31) - 1) // d * d - 1]
        pts += [rng.randrange(0, 1 << 31) for _ in range(50)] + [k * d + e for k in (3, 1000, (1 << 31) // d - 2) for e in (-1, 0, 1)]
        for n in pts:
            if 0 <= n < (1 << 31):
                assert ((2 * n * mg.value) >> 32) >> sh.value == n // d, (n, d)
    assert lib.hdy_fastdiv_magic(0, ctypes.byref(mg), ctypes.byref(sh)) < 0


def test_launch_list_executor_table_and_malformed_programs(lib):
    """hdy_exec_run (csrc/exec.hip): every entry point of the header whose last parameter is the stream can be listed, nothing else can; a
    malformed program is a status (decided on the host, before any launch), not a crash."""
    import ctypes
    text = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', 'hdyolo.h')).read(), flags=re.S)
    decls = re.findall(r'\b(hdy_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;', text, flags=re.S)
    assert len(decls) >= 80
    for name, params in decls:
        takes_stream = re.search(r'void\s*\*\s*stream\s*$', params.strip()) is not None
        if name in ('hdy_exec_run', 'hdy_exec_join'):          # the executor itself is not a list item
            takes_stream = False
        assert (lib.hdy_exec_op(name.encode()) >= 0) == takes_stream, name
        if takes_stream:
            assert _lib.SIGNATURES[name][0] is ctypes.c_int and _lib.SIGNATURES[name][1][-1] is ctypes.c_void_p, name
    assert lib.hdy_exec_op(b'hdy_no_such_entry') == -1

    def run(words):
        arr = (ctypes.c_ulonglong * max(1, len(words)))(*words)
        return lib.hdy_exec_run(arr, len(words), None, None)

    assert run([]) == _lib.OK
    op = lib.hdy_exec_op(b'hdy_scale_inplace')
    assert run([op]) == _lib.EINVAL and b'truncated' in lib.hdy_last_error()
    assert run([op, 9, 0, 0]) == _lib.EINVAL                                   # more arguments than words
    assert run([op, 2, 0, 0]) == _lib.EINVAL and b'hdy_scale_inplace takes 4' in lib.hdy_last_error()
    assert run([12345, 0]) == _lib.EINVAL and b'unknown op' in lib.hdy_last_error()
    assert run([0xF0F0F0F0, 2, 1, 0]) == _lib.EINVAL                           # a fork without a side stream
    assert run([0xF0F0F0F1, 2, 1, 1]) == _lib.EINVAL                           # a join takes one argument
    assert run([op, 4, 0, 16, 0, _lib.BF16]) == _lib.EINVAL                    # well formed: the entry point's own argument check answers (null pointer)

    def run_side(words):                                                       # with a (never used) side stream: the host-side checks come first
        arr = (ctypes.c_ulonglong * len(words))(*words)
        return lib.hdy_exec_run(arr, len(words), None, ctypes.c_void_p(0x10))

    # a fork whose length word is huge: `i + length` would wrap in 64 bits and pass a naive bound check
    assert run_side([0xF0F0F0F0, 2, 1, (1 << 64) - 1]) == _lib.EINVAL and b'bad fork at word 0' in lib.hdy_last_error()
    assert run_side([0xF0F0F0F0, 2, 1, 7, op, 4, 0, 16, 0, _lib.BF16]) == _lib.EINVAL and b'bad fork at word 0' in lib.hdy_last_error()   # one word longer than the program
    assert run_side([0xF0F0F0F1, 2, 1, 1]) == _lib.EINVAL and b'bad join at word 0' in lib.hdy_last_error()                  # messages name the item's first word

def test_program_words_follow_the_record_list(lib):
    """ops.Program (the Python side of hdy_exec_run) on a hand-made record list, no GPU: one segment per stretch between host callbacks, fork bodies
    inline behind their length, arguments widened by the ctypes type of the parameter (negative ints, floats by bit pattern, None -> 0)."""
    import struct
    from hd_yolo_amd import ops

    class Side:                                    # stands in for ops.SideStream (never run here)
        pass

    side = Side()
    scale = ('hdy_scale_inplace', (0x1000, 7, 0x2000, _lib.BF16), ())
    addi = ('hdy_add_inplace', (0x3000, 64, 0x4000, 64, -5, 64, _lib.BF16), ())
    fin = ('hdy_bn_finalize_sums', (0x10, 8, 0x20, 8, 8) + (None,) * 8 + (1e-3, 0.03) + (0x30,) * 4, ())
    seen = []
    prog = ops.Program([scale, ('@fork', side, [addi, scale], 9), ('@call', lambda: seen.append(1)), fin, ('@join', side, 9)])
    assert [s[0] for s in prog.segments] == ['words', 'call', 'words'] and prog.side is side
    w0, w1 = list(prog.segments[0][1]), list(prog.segments[2][1])
    op_s, op_a, op_f = (lib.hdy_exec_op(n) for n in (b'hdy_scale_inplace', b'hdy_add_inplace', b'hdy_bn_finalize_sums'))
    item_s = [op_s, 4, 0x1000, 7, 0x2000, _lib.BF16]
    item_a = [op_a, 7, 0x3000, 64, 0x4000, 64, (1 << 64) - 5, 64, _lib.BF16]
    tb = prog.token_base                                                   # each program owns a range of the library's fork tokens
    keep = ops.Program([scale])
    assert keep.token_base == tb + 10                                      # a live program's range is its own ...
    del keep
    assert ops.Program([scale]).token_base == tb + 10                      # ... and returns to the table with it
    assert w0 == item_s + [ops.EXEC_FORK, 2, tb + 9, len(item_a) + len(item_s)] + item_a + item_s
    f32 = lambda v: struct.unpack('<I', struct.pack('<f', v))[0]
    assert w1 == [op_f, 19, 0x10, 8, 0x20, 8, 8] + [0] * 8 + [f32(1e-3), f32(0.03)] + [0x30] * 4 + [ops.EXEC_JOIN, 1, tb + 9]
    with pytest.raises(_lib.HdyError, match='cannot be listed'):
        ops.Program([('hdy_scale_inplace', (1, 2, 3), ())])              # one argument short


def test_c128_swizzle_is_conflict_free():
    """conv3x3_c128.hip keeps its patch as 256-byte pixel rows (all 64 LDS banks once per row), so the 16 lanes of every ds_read_b128 lane group
    (MI355X_MICROARCH.md, LDS table: {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32) must read 16 DIFFERENT 16-byte chunks.  Lane (fr, fq) of
    a fragment reads patch column (fr & 7) + s (s = 0, 1, 2: the filter column) and logical chunk ks * 4 + fq; the kernel XORs the chunk with
    (column & 7) << 1.  Restated here so that a change of the key or of the lane -> pixel map cannot go unnoticed on a box without a GPU."""
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
    groups += [[l + 32 for l in g] for g in groups]
    for s in range(3):
        for ks in range(4):
            for g in groups:
                chunks = set()
                for lane in g:
                    fr, fq = lane & 15, lane >> 4
                    col = (fr & 7) + s
                    chunks.add((ks * 4 + fq) ^ ((col & 7) << 1))
                assert len(chunks) == 16, (s, ks, g, sorted(chunks))
