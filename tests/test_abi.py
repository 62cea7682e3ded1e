"""CPU tests of the boundary: libh2g.so loads, exports every symbol include/h2g.h declares, its structs have
the layout the ctypes mirror assumes, and it refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from hisat2_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "h2g.h")


def _declared(hdr=HDR):
    txt = open(hdr).read()
    return re.findall(r"H2G_EXPORT\s+[\w\s\*]+?\b(h2g_\w+)\s*\(", txt)


def test_library_exports_the_sam_emitter():
    """include/h2g_sam.h (host-side SAM emission) lives in the same library"""
    names = _declared(os.path.join(ROOT, "include", "h2g_sam.h"))
    assert sorted(names) == ["h2g_sam_add_read_group", "h2g_sam_add_splice_sites", "h2g_sam_close", "h2g_sam_collect_novel_sites", "h2g_sam_format_paired", "h2g_sam_format_paired_compact", "h2g_sam_format_paired_dense", "h2g_sam_format_unpaired", "h2g_sam_format_unpaired_compact", "h2g_sam_format_unpaired_dense", "h2g_sam_header", "h2g_sam_novel_splice_sites_text", "h2g_sam_open", "h2g_sam_read_splice_site_file", "h2g_sam_set_chrname_mode", "h2g_sam_set_first_read_id", "h2g_sam_set_header_options", "h2g_sam_set_long_edits", "h2g_sam_set_new_summary", "h2g_sam_set_no_unal", "h2g_sam_set_report_policy", "h2g_sam_set_rna_strandness", "h2g_sam_set_score_min", "h2g_sam_set_secondary", "h2g_sam_set_splice_sites", "h2g_sam_set_templatelen_adjustment", "h2g_sam_set_threads", "h2g_sam_summary", "h2g_sam_take_novel_sites"]
    L = api.lib()
    for n in names:
        assert hasattr(L, n), n


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 18
    L = api.lib()
    for n in names:
        assert hasattr(L, n), n
    assert sorted(names) == sorted(api.EXPORTS)


def test_struct_layouts_match_header(tmp_path):
    structs = {"h2g_load_opts": api.LoadOpts, "h2g_index_info": api.IndexInfo, "h2g_fm_query": api.FmQuery,
               "h2g_fm_hit": api.FmHit, "h2g_sa_query": api.SaQuery, "h2g_coord": api.Coord,
               "h2g_sa_result": api.SaResult, "h2g_edit": api.Edit, "h2g_ghit": api.GHit, "h2g_ext_args": api.ExtArgs,
               "h2g_ext_result": api.ExtResult, "h2g_seed_result": api.SeedResult, "h2g_seed_params": api.SeedParams,
               "h2g_counters": api.Counters, "h2g_alnres": api.AlnRes, "h2g_read_result": api.ReadResult,
               "h2g_align_params": api.AlignParams, "h2g_pair_result": api.PairResult}
    src = tmp_path / "sz.c"
    body = "".join(f'printf("{n} %zu\\n", sizeof({n}));' for n in structs)
    src.write_text(f'#include <stdio.h>\n#include "{HDR}"\nint main(void){{{body}return 0;}}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-std=c11", "-o", str(exe), str(src)], check=True)   # header is plain C
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    for line in out.splitlines():
        n, sz = line.split()
        assert C.sizeof(structs[n]) == int(sz), n


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_have_gpu(), reason="GPU present")
def test_no_cpu_fallback(g1_index):
    """Without a HIP device every entry point fails loudly instead of computing on the host."""
    with pytest.raises(api.H2GError) as e:
        api.Index(g1_index)
    assert "-3" in str(e.value)   # H2G_ERR_DEVICE
    with pytest.raises(api.H2GError):
        api.Index(synth_sides=1024)


def test_stream_defaults_are_what_the_python_mirror_says():
    """H2G_DEFAULT_TAIL / H2G_DEFAULT_ALIGN_MATE of csrc/h2g_kernels.hip (what go_run does without H2G_FAST_TAIL / H2G_FAST_AM) == api.DEFAULT_*,
    which bench.py names in its line"""
    import re
    from hisat2_amd import api
    src = open(os.path.join(ROOT, "hisat2_amd", "csrc", "h2g_kernels.hip")).read()
    assert int(re.search(r"#define H2G_DEFAULT_TAIL (\d+)", src).group(1)) == api.DEFAULT_TAIL
    assert int(re.search(r"#define H2G_DEFAULT_ALIGN_MATE (\d+)", src).group(1)) == api.DEFAULT_ALIGN_MATE


def test_pmc_record_is_attached_only_to_the_sources_it_was_taken_on():
    """The newest profiles/rNN_pmc_traffic.json (rocprofv3 FETCH_SIZE / WRITE_SIZE of the headline kernel at GRCh38 size) carries the hash of the kernel
    sources it was taken on; bench.py attaches `roofline.traffic` only while that equals the hash of the sources in the tree (comments and blank space
    apart) and the workload is the record's.  A kernel change after the profile yields `traffic: null` + a note — never a stale figure in the bench line."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    path, rec = bench.newest_pmc_record()
    assert rec is not None and rec["kernel"].startswith("k_go_fast")
    fresh = rec["kernel_sources_sha16"] == bench.kernel_sources_sha16()
    r = {"traffic": None}
    assert bench.attach_pmc_traffic(r, rec["pairs_per_launch"], rec["genome"], True) == fresh
    if fresh:
        assert r["traffic"] == rec["traffic_bytes_per_launch"] and r["traffic_record"].startswith("profiles/")
    else:
        assert r["traffic"] is None and "not attached" in r["traffic_note"]
    r = {"traffic": None}                                           # another workload: never attached
    assert not bench.attach_pmc_traffic(r, rec["pairs_per_launch"] // 2, rec["genome"], True) and r["traffic"] is None
    assert bench._strip_comments('a = "//x"; // c\n/* d */ b /* e\n f */ c\n\n') == 'a = "//x";\nb   c'
