// h2g_cli.cpp — `hisat2-align-amd`: the reference's `hisat2-align-s -x <index> -U/-1/-2 … -S out.sam` command line for the
// part of HISAT2 that is built here (--no-spliced-alignment; linear or SNP-graph index; unpaired or paired reads).
// Host code only: batched read ingestion (FASTA / FASTQ, the parse rules of pat.cpp:725-1010), the C ABI of
// include/h2g.h for HI_Aligner::go on the GPU, include/h2g_sam.h for the sink + SAM text (N1).  There is no CPU aligner in
// here: without a GPU h2g_index_load fails and so does this program.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <thread>
#include <chrono>
#include "../../include/h2g.h"
#include "../../include/h2g_sam.h"

namespace {

struct Batch {
	std::vector<uint8_t>  codes;
	std::vector<uint32_t> offs, noffs;
	std::string           quals, names;
	bool                  have_quals = false;
	size_t n() const { return offs.empty() ? 0 : offs.size() - 1; }
	void clear() { codes.clear(); offs.assign(1, 0); noffs.assign(1, 0); quals.clear(); names.clear(); }
};

// asc2dnacat > 0 (alphabet.cpp:36-58): DNA letters, IUPAC codes, N and '-' are read characters; asc2dna (alphabet.cpp:298)
inline bool is_read_char(int c) {
	switch(c | 0x20) { case 'a': case 'b': case 'c': case 'd': case 'g': case 'h': case 'k': case 'm': case 'n': case 'r': case 's': case 't':
	                   case 'v': case 'w': case 'x': case 'y': return true; }
	return c == '-';
}
inline uint8_t base_code(int c) { switch(c | 0x20) { case 'c': return 1; case 'g': return 2; case 't': return 3; case 'n': return 4; } return 0; }

class Reader {          // sequential reader over a list of FASTA / FASTQ files (pat.cpp FastaPatternSource / FastqPatternSource)
public:
	Reader(const std::vector<std::string>& files, bool fasta) : files_(files), fasta_(fasta) {}
	~Reader() { if(f_) fclose(f_); }
	// appends up to `max` reads to b; returns the number appended (0 at end of input)
	size_t fill(Batch& b, size_t max) {
		size_t got = 0;
		std::string name, seq, qual, line;
		while(got < max) {
			if(!next_record(name, seq, qual)) break;
			if(name.empty()) name = std::to_string(count_);
			count_++;
			b.names += name; b.noffs.push_back((uint32_t)b.names.size());
			for(char c : seq) b.codes.push_back(base_code(c));
			b.offs.push_back((uint32_t)b.codes.size());
			if(!fasta_) { b.have_quals = true; b.quals += qual; }
			got++;
		}
		return got;
	}
private:
	bool getline_(std::string& s) {
		s.clear();
		for(;;) {
			if(!f_) { if(fi_ >= files_.size()) return false; f_ = fopen(files_[fi_].c_str(), "rb"); if(!f_) { fprintf(stderr, "Error: could not open %s\n", files_[fi_].c_str()); exit(1); } fi_++; }
			int c;
			bool any = false;
			while((c = getc_unlocked(f_)) != EOF) { any = true; if(c == '\n') break; if(c != '\r') s.push_back((char)c); }
			if(c == EOF && !any) { fclose(f_); f_ = nullptr; continue; }
			return true;
		}
	}
	bool next_record(std::string& name, std::string& seq, std::string& qual) {
		std::string line;
		name.clear(); seq.clear(); qual.clear();
		if(fasta_) {
			if(pending_.empty()) { do { if(!getline_(line)) return false; } while(line.empty() || line[0] == '#' || line[0] == ';'); }
			else { line = pending_; pending_.clear(); }
			if(line[0] != '>') { fprintf(stderr, "Error: reads file does not look like a FASTA file\n"); exit(1); }
			name = line.substr(1);
			while(getline_(line)) {
				if(!line.empty() && line[0] == '>') { pending_ = line; break; }
				for(char c : line) if(is_read_char((unsigned char)c)) seq.push_back(c);
			}
			return true;
		}
		do { if(!getline_(line)) return false; } while(line.empty());
		if(line[0] != '@') { fprintf(stderr, "Error: reads file does not look like a FASTQ file\n"); exit(1); }
		name = line.substr(1);
		if(!getline_(line)) return false;
		for(char c : line) { if(c == '.') c = 'N'; if(is_read_char((unsigned char)c)) seq.push_back(c); }
		if(!getline_(line)) return false;      // '+' line
		if(!getline_(qual)) return false;
		if(qual.size() < seq.size()) { fprintf(stderr, "Error: Read %s has more read characters than quality values.\n", name.c_str()); exit(1); }
		qual.resize(seq.size());
		return true;
	}
	std::vector<std::string> files_;
	bool fasta_;
	size_t fi_ = 0;
	FILE* f_ = nullptr;
	std::string pending_;
	uint64_t count_ = 0;
};

std::vector<std::string> split_commas(const char* s) {
	std::vector<std::string> v;
	std::string cur;
	for(; *s; s++) { if(*s == ',') { if(!cur.empty()) v.push_back(cur); cur.clear(); } else cur.push_back(*s); }
	if(!cur.empty()) v.push_back(cur);
	return v;
}
void die(const char* what) { fprintf(stderr, "hisat2-align-amd: %s (%s)\n", what, h2g_last_error()); exit(1); }
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

int main(int argc, char** argv) {
	std::string base, outfn;
	std::vector<std::string> u, m1, m2;
	bool fasta = false, nospliced = false, nohead = false;
	uint32_t dp = 0;
	size_t batch = 1u << 20;
	int device = 0;
	std::string cmdline;
	for(int i = 0; i < argc; i++) { if(i) cmdline.push_back(' '); cmdline += argv[i]; }
	for(int i = 1; i < argc; i++) {
		const std::string a = argv[i];
		auto need = [&](const char* o) { if(i + 1 >= argc) { fprintf(stderr, "option %s needs an argument\n", o); exit(1); } return argv[++i]; };
		if(a == "-x") base = need("-x");
		else if(a == "-U") { auto v = split_commas(need("-U")); u.insert(u.end(), v.begin(), v.end()); }
		else if(a == "-1") { auto v = split_commas(need("-1")); m1.insert(m1.end(), v.begin(), v.end()); }
		else if(a == "-2") { auto v = split_commas(need("-2")); m2.insert(m2.end(), v.begin(), v.end()); }
		else if(a == "-S") outfn = need("-S");
		else if(a == "-f") fasta = true;
		else if(a == "-q") fasta = false;
		else if(a == "-p" || a == "--threads") need("-p");                       // host threads: the device does the work
		else if(a == "--no-spliced-alignment") nospliced = true;
		else if(a == "--bowtie2-dp") dp = (uint32_t)atoi(need("--bowtie2-dp"));
		else if(a == "--no-hd" || a == "--no-head") nohead = true;
		else if(a == "--batch") batch = (size_t)atoll(need("--batch"));
		else if(a == "--device") device = atoi(need("--device"));
		else { fprintf(stderr, "hisat2-align-amd: option %s is not built (see DESIGN.md, scope)\n", a.c_str()); return 1; }
	}
	if(base.empty() || (u.empty() && (m1.empty() || m2.empty()))) {
		fprintf(stderr, "usage: hisat2-align-amd -x <ht2-base> {-U <r.fq> | -1 <m1.fq> -2 <m2.fq>} [-f|-q] --no-spliced-alignment [--bowtie2-dp 0|1|2] [-S out.sam]\n");
		return 1;
	}
	if(!nospliced) { fprintf(stderr, "hisat2-align-amd: spliced alignment is not built yet; pass --no-spliced-alignment\n"); return 1; }
	const bool paired = u.empty();
	const double t0 = now();
	h2g_load_opts lo; h2g_load_opts_init(&lo); lo.device = device; lo.load_local = 1;
	h2g_index* ix = nullptr;
	if(h2g_index_load(base.c_str(), &lo, &ix) != H2G_OK) die("cannot load the index onto the GPU");
	h2g_sam* sam = nullptr;
	if(h2g_sam_open(base.c_str(), &sam) != H2G_OK) die("cannot read reference names");
	h2g_align_params P; h2g_align_params_init(&P, ix);
	P.bowtie2_dp = dp;
	FILE* out = outfn.empty() ? stdout : fopen(outfn.c_str(), "wb");
	if(!out) { fprintf(stderr, "cannot open %s\n", outfn.c_str()); return 1; }
	std::vector<char> buf(1 << 20);
	if(!nohead) {
		const size_t need = h2g_sam_header(sam, cmdline.c_str(), nullptr, 0);
		buf.resize(need + 1);
		h2g_sam_header(sam, cmdline.c_str(), buf.data(), buf.size());
		fwrite(buf.data(), 1, need, out);
	}
	const double t1 = now();
	Reader ra(paired ? m1 : u, fasta), rb(m2, fasta);
	h2g_stream* st = nullptr;
	Batch A[2], B[2];                                 // double buffer: batch k+1 is parsed while batch k is on the GPU
	uint64_t nreads = 0, naligned = 0, novf = 0;
	double t_gpu = 0, t_fmt = 0, t_parse = 0;
	size_t stream_reads = 0, stream_bases = 0;
	int cur = 0;
	A[0].clear(); B[0].clear();
	double tp = now();
	size_t n = ra.fill(A[0], batch);
	if(paired && rb.fill(B[0], n) != n) { fprintf(stderr, "Error, fewer reads in file specified with -2 than in file specified with -1\n"); return 1; }
	t_parse += now() - tp;
	std::vector<h2g_read_result> res;
	std::vector<h2g_pair_result> pres;
	std::vector<h2g_alnres> aln, aln2;
	while(n > 0) {
		Batch& a = A[cur]; Batch& b = B[cur];
		size_t bases = a.codes.size();
		if(paired && b.codes.size() > bases) bases = b.codes.size();
		if(!st || n > stream_reads || bases > stream_bases) {
			if(st) h2g_stream_free(st);
			stream_reads = n > batch ? n : batch; stream_bases = bases + bases / 4 + 1024;
			if(h2g_stream_create(ix, stream_reads, stream_bases, &st) != H2G_OK) die("cannot create the device stream");
		}
		const double tg = now();
		if(h2g_set_reads(st, a.codes.data(), a.offs.data(), a.have_quals ? a.quals.data() : nullptr, n) != H2G_OK) die("h2g_set_reads");
		if(h2g_set_read_names(st, a.names.data(), a.noffs.data(), n) != H2G_OK) die("h2g_set_read_names");
		if(paired) {
			if(h2g_set_mates(st, b.codes.data(), b.offs.data(), b.have_quals ? b.quals.data() : nullptr, b.names.data(), b.noffs.data(), n) != H2G_OK) die("h2g_set_mates");
			if(h2g_align_pairs_run(st, &P) != H2G_OK) die("h2g_align_pairs_run");
		} else if(h2g_align_run(st, &P) != H2G_OK) die("h2g_align_run");
		// ingest the next batch on this thread while the kernel runs (the run calls are asynchronous)
		const int nxt = cur ^ 1;
		A[nxt].clear(); B[nxt].clear();
		tp = now();
		const size_t n2 = ra.fill(A[nxt], batch);
		if(paired && rb.fill(B[nxt], n2) != n2) { fprintf(stderr, "Error, fewer reads in file specified with -2 than in file specified with -1\n"); return 1; }
		t_parse += now() - tp;
		size_t used = 0;
		if(paired) {
			pres.resize(n); aln.resize(n * H2G_PAIR_RES_CAP); aln2.resize(n * H2G_PAIR_RES_CAP);
			if(h2g_align_pairs_fetch(st, pres.data(), aln.data(), aln2.data(), 0, n) != H2G_OK) die("h2g_align_pairs_fetch");
			t_gpu += now() - tg;
			const double tf = now();
			buf.resize(n * 1400 + 6 * (a.codes.size() + b.codes.size()) + 4096);
			h2g_status rc = h2g_sam_format_paired(sam, a.codes.data(), a.offs.data(), a.have_quals ? a.quals.data() : nullptr, a.names.data(), a.noffs.data(),
			                                      b.codes.data(), b.offs.data(), b.have_quals ? b.quals.data() : nullptr, b.names.data(), b.noffs.data(), n,
			                                      pres.data(), aln.data(), aln2.data(), P.khits, buf.data(), buf.size(), &used);
			if(rc != H2G_OK) {
				buf.resize(used + 16);
				rc = h2g_sam_format_paired(sam, a.codes.data(), a.offs.data(), a.have_quals ? a.quals.data() : nullptr, a.names.data(), a.noffs.data(),
				                           b.codes.data(), b.offs.data(), b.have_quals ? b.quals.data() : nullptr, b.names.data(), b.noffs.data(), n,
				                           pres.data(), aln.data(), aln2.data(), P.khits, buf.data(), buf.size(), &used);
				if(rc != H2G_OK) die("h2g_sam_format_paired");
			}
			for(size_t i = 0; i < n; i++) { naligned += pres[i].npairs > 0; novf += pres[i].overflow != 0; }
			t_fmt += now() - tf;
		} else {
			res.resize(n); aln.resize(n * H2G_ALN_CAP);
			if(h2g_align_fetch(st, res.data(), aln.data(), 0, n) != H2G_OK) die("h2g_align_fetch");
			t_gpu += now() - tg;
			const double tf = now();
			buf.resize(n * 700 + 3 * a.codes.size() + 4096);
			h2g_status rc = h2g_sam_format_unpaired(sam, a.codes.data(), a.offs.data(), a.have_quals ? a.quals.data() : nullptr, a.names.data(), a.noffs.data(), n,
			                                        res.data(), aln.data(), buf.data(), buf.size(), &used);
			if(rc != H2G_OK) {
				buf.resize(used + 16);
				rc = h2g_sam_format_unpaired(sam, a.codes.data(), a.offs.data(), a.have_quals ? a.quals.data() : nullptr, a.names.data(), a.noffs.data(), n,
				                             res.data(), aln.data(), buf.data(), buf.size(), &used);
				if(rc != H2G_OK) die("h2g_sam_format_unpaired");
			}
			for(size_t i = 0; i < n; i++) { naligned += res[i].nselect > 0; novf += res[i].overflow != 0; }
			t_fmt += now() - tf;
		}
		fwrite(buf.data(), 1, used, out);
		nreads += n;
		n = n2;
		cur = nxt;
	}
	if(out != stdout) fclose(out);
	const double t2 = now();
	fprintf(stderr, "%llu %s; %llu %s (%.2f%%)\n", (unsigned long long)nreads, paired ? "pairs" : "reads", (unsigned long long)naligned,
	        paired ? "aligned concordantly at least once" : "aligned", nreads ? 100.0 * naligned / nreads : 0.0);
	if(novf) fprintf(stderr, "Warning: %llu %s exceeded a fixed device capacity (h2g overflow bit); rerun them with the reference aligner\n",
	                 (unsigned long long)novf, paired ? "pairs" : "reads");
	fprintf(stderr, "time: index load %.2f s, align+fetch %.2f s (includes overlapped parsing %.2f s), SAM formatting %.2f s, total %.2f s\n", t1 - t0, t_gpu,
	        t_parse, t_fmt, t2 - t0);
	if(st) h2g_stream_free(st);
	h2g_sam_close(sam);
	h2g_index_free(ix);
	return 0;
}
