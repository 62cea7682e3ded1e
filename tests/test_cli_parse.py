"""N2: the command line's parallel FASTA / FASTQ reader (hisat2_amd/csrc/h2g_cli.cpp) — same reads, names and qualities for
every thread count and batch size, CRLF and multi-line FASTA included (parse rules of pat.cpp:725-1010).  No GPU needed."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "hisat2_amd", "hisat2-align-amd")
pytestmark = pytest.mark.skipif(not os.path.exists(CLI), reason="hisat2-align-amd not built")


def fnv(chunks):
    h = 1469598103934665603
    for c in chunks:
        for b in c:
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def expected(names, seqs, quals, batch):
    code = {"A": 0, "C": 1, "G": 2, "T": 3, "N": 4}
    n = len(names)
    h = 1469598103934665603
    tot = 0
    for b0 in range(0, n, batch):
        sl = slice(b0, min(n, b0 + batch))
        codes = bytes(code.get(c.upper(), 0) for s in seqs[sl] for c in s)
        nm = "".join(names[sl]).encode()
        q = "".join(quals[sl]).encode() if quals else b""
        lens = b"".join(np.uint32(len(s)).tobytes() + np.uint32(len(m)).tobytes() for s, m in zip(seqs[sl], names[sl]))
        for c in (codes, nm, q, lens):
            for b in c:
                h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        tot += len(codes)
    return n, tot, h


@pytest.mark.parametrize("fmt", ["fasta", "fasta_multiline_crlf", "fastq"])
def test_reader_is_thread_and_batch_invariant(tmp_path, fmt):
    rng = np.random.default_rng(5)
    n = 3000
    names = [("r%d extra words" % i) if i % 3 else "" for i in range(n)]          # empty name -> read id (pat.cpp:842)
    seqs = ["".join(rng.choice(list("ACGTNacgtRY"), size=int(rng.integers(30, 160)))) for _ in range(n)]
    quals = ["".join(chr(int(q)) for q in rng.integers(35, 74, size=len(s))) for s in seqs] if fmt == "fastq" else None
    path = tmp_path / ("r.fq" if fmt == "fastq" else "r.fa")
    with open(path, "w", newline="") as f:
        for i in range(n):
            if fmt == "fastq":
                f.write("@%s\n%s\n+\n%s\n" % (names[i], seqs[i], quals[i]))
            elif fmt == "fasta":
                f.write(">%s\n%s\n" % (names[i], seqs[i]))
            else:
                s = seqs[i]
                f.write(">%s\r\n%s\r\n%s\r\n" % (names[i], s[:40], s[40:]))
    names = [nm if nm else str(i) for i, nm in enumerate(names)]
    seqs2 = ["".join(c if c.upper() in "ACGTN" else "A" for c in s) for s in seqs]     # asc2dna: IUPAC codes read as 0
    want = None
    for threads, batch in ((1, 1 << 20), (4, 1 << 20), (7, 700), (3, 1)):
        out = subprocess.run([CLI, "--parse-only", "-f" if fmt != "fastq" else "-q", "-U", str(path), "-p", str(threads), "--batch", str(batch),
                              "-x", "unused"], check=True, capture_output=True, text=True).stdout.split()
        got = (int(out[0]), int(out[1]), int(out[2], 16))
        assert got == expected(names, seqs2, quals, batch), (threads, batch)
    import gzip, shutil
    with open(path, "rb") as fi, gzip.open(str(path) + ".gz", "wb") as fo:       # gzipped input gives the same reads
        shutil.copyfileobj(fi, fo)
    out = subprocess.run([CLI, "--parse-only", "-f" if fmt != "fastq" else "-q", "-U", str(path) + ".gz", "-p", "4", "-x", "unused"], check=True,
                         capture_output=True, text=True).stdout.split()
    assert (int(out[0]), int(out[1]), int(out[2], 16)) == expected(names, seqs2, quals, 1 << 20)


def test_phred64_input(tmp_path):
    """--phred64: a FASTQ file with 64-based qualities parses to what its 33-based twin parses to (charToPhred33 qual.h:126)"""
    import numpy as np
    rng = np.random.default_rng(5)
    p33, p64 = tmp_path / "a33.fq", tmp_path / "a64.fq"
    with open(p33, "w") as f33, open(p64, "w") as f64:
        for i in range(300):
            L = int(rng.integers(30, 120))
            seq = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=L))
            q = rng.integers(0, 41, size=L)
            f33.write(f"@r{i}\n{seq}\n+\n" + "".join(chr(33 + int(x)) for x in q) + "\n")
            f64.write(f"@r{i}\n{seq}\n+\n" + "".join(chr(64 + int(x)) for x in q) + "\n")
    run = lambda path, extra: subprocess.run([CLI, "--parse-only", "-q", "-U", str(path), "-x", "unused"] + extra, check=True, stdout=subprocess.PIPE).stdout
    assert run(p64, ["--phred64"]) == run(p33, [])
    assert run(p64, []) != run(p33, [])
    assert subprocess.run([CLI, "--parse-only", "-q", "-U", str(p33), "-x", "unused", "--phred64"], stdout=subprocess.PIPE, stderr=subprocess.PIPE).returncode != 0


def test_raw_and_command_line_reads(tmp_path):
    """-r (one sequence per line, blank lines skipped) and -c (comma-separated sequences) parse to what the FASTA file with empty
    record names parses to: reads numbered 0, 1, … , qualities 'I'"""
    seqs = ["ACGTACGTACGTAGCTAGCTAGCATCGATCGATCGTAGCTAGCTAG", "GGGGACGTNNACGTAGCTAGCTAGCATCGATCGATCGTAGCTAGCTAG", "TTTTGGGGCCCCAAAA"]
    fa, raw = tmp_path / "e.fa", tmp_path / "e.txt"
    fa.write_text("".join(f">\n{s}\n" for s in seqs))
    raw.write_text(seqs[0] + "\n\n" + seqs[1] + "\r\n" + seqs[2])
    run = lambda args: subprocess.run([CLI, "--parse-only", "-x", "unused"] + args, check=True, stdout=subprocess.PIPE).stdout
    want = run(["-f", "-U", str(fa)])
    assert run(["-r", "-U", str(raw)]) == want
    assert run(["-c", "-U", ",".join(seqs)]) == want


def test_absurd_splice_site_window_is_refused_by_name(tmp_path):
    """in the temporary-splice-site mode a wave (the window, or 1000 x -p) sizes the streams and the result rows whatever --batch says; a window beyond
    what one resident batch can be is refused with an error that names --ss-window and --batch — before any device or index is touched (no GPU needed)."""
    r = tmp_path / "r.fa"
    r.write_text(">0\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n")
    p = subprocess.run([CLI, "-f", "-x", str(tmp_path / "no_such_index"), "-U", str(r), "--ss-window", "5000000", "-S", str(tmp_path / "o.sam")], capture_output=True, text=True)
    assert p.returncode == 1
    assert "--ss-window" in p.stderr and "--batch" in p.stderr and "4194304" in p.stderr
