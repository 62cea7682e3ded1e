"""The request-site table of the go() machine (H2G_MACH_SITES in hisat2_amd/csrc/h2g_machine.h) lists exactly the (primitive,
resume pc) pairs the machine uses: the kernels queue reads per site, and a site missing from the table would have no queue."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_site_table_matches_the_request_sites():
    src = open(os.path.join(ROOT, "hisat2_amd", "csrc", "h2g_machine.h")).read()
    table_src = src[src.index("#define H2G_MACH_SITES_DB(X)"):src.index("enum : uint32_t {\n#define X(OPC, PC) SITE_##PC")]   # database sites (H2G_SPLICE_DB) + the rest
    table = re.findall(r"X\((OP_[A-Z]+), (PC_[A-Z0-9_]+)\)", table_src)
    body = src[src.index("void mach_step("):]
    used = set(re.findall(r"M_OP\((OP_[A-Z]+), (PC_[A-Z0-9_]+)\)", body))
    assert len(table) == len(set(table))
    assert set(table) == used
    assert len({pc for _, pc in table}) == len(table)      # a resume pc belongs to one site: pc -> ring is a function
    assert len(table) + 1 <= 64                            # ring census: one lane per ring
