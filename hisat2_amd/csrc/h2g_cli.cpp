// h2g_cli.cpp — `hisat2-align-amd`: the reference's `hisat2-align-s -x <index> -U/-1/-2 … -S out.sam` command line for the
// part of HISAT2 that is built here (--no-spliced-alignment; linear or SNP-graph index; unpaired or paired reads).
// Host code only: batched read ingestion (FASTA / FASTQ, the parse rules of pat.cpp:725-1010), the C ABI of
// include/h2g.h for HI_Aligner::go on the GPU, include/h2g_sam.h for the sink + SAM text (N1).  There is no CPU aligner in
// here: without a GPU h2g_index_load fails and so does this program.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <string>
#include <map>
#include <array>
#include <vector>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <deque>
#include <chrono>
#include <algorithm>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <zlib.h>
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
// the two per-character tests as tables (0xff = not a base of the record): FASTA keeps is_read_char() characters, FASTQ keeps '.' (as N) and every isalpha() character
struct BaseTables {
	uint8_t fa[256], fq[256];
	BaseTables() {
		for(int c = 0; c < 256; c++) {
			fa[c] = is_read_char(c) ? base_code(c) : 0xff;
			const int d = c == '.' ? 'N' : c;
			fq[c] = isalpha(d) ? base_code(d) : 0xff;
		}
	}
};
inline const BaseTables& base_tables() { static const BaseTables t; return t; }
// appends the bases of [q, e) to `codes` through table `tb`: written unconditionally, kept when they are bases (no branch per character, no push_back)
inline void append_bases(std::vector<uint8_t>& codes, const char* q, const char* e, const uint8_t* tb) {
	const size_t at = codes.size();
	codes.resize(at + (size_t)(e - q));
	uint8_t* o = codes.data() + at;
	for(; q < e; q++) { const uint8_t v = tb[(unsigned char)*q]; *o = v; o += v != 0xff; }
	codes.resize((size_t)(o - codes.data()));
}

// Sequential stream of reads over a list of FASTA / FASTQ files (pat.cpp FastaPatternSource / FastqPatternSource), parsed
// in parallel: a file is mapped, the record starts are found by all threads (FASTA: lines beginning with '>'; FASTQ: every
// fourth line), and each fill() hands contiguous record ranges to the threads and concatenates their output in file order.
class Reader {
public:
	Reader(const std::vector<std::string>& files, bool fasta, int threads, uint32_t trim5 = 0, uint32_t trim3 = 0)
		: files_(files), fasta_(fasta), T_(threads < 1 ? 1 : threads), trim5_(trim5), trim3_(trim3) {}
	~Reader() { unmap(); }
	size_t fill(Batch& b, size_t max) {
		size_t got = 0;
		while(got < max) {
			if(cur_ >= nrec()) { if(!next_file()) break; continue; }
			const size_t take = std::min(max - got, nrec() - cur_);
			const size_t T = std::min<size_t>((size_t)T_, take / 4096 + 1);
			std::vector<Batch> part(T);
			auto work = [&](size_t t) {
				Batch& pb = part[t];
				pb.clear();
				const size_t rb = cur_ + take * t / T, re = cur_ + take * (t + 1) / T;
				for(size_t r = rb; r < re; r++) parse_record(r, pb);
			};
			std::vector<std::thread> th;
			for(size_t t = 1; t < T; t++) th.emplace_back(work, t);
			work(0);
			for(auto& x : th) x.join();
			for(size_t t = 0; t < T; t++) {
				const Batch& pb = part[t];
				const uint32_t cb = (uint32_t)b.codes.size(), nb = (uint32_t)b.names.size();
				b.codes.insert(b.codes.end(), pb.codes.begin(), pb.codes.end());
				b.names += pb.names;
				if(!fasta_) { b.have_quals = true; b.quals += pb.quals; }
				for(size_t k = 1; k < pb.offs.size(); k++) { b.offs.push_back(cb + pb.offs[k]); b.noffs.push_back(nb + pb.noffs[k]); }
			}
			cur_ += take; got += take; count_ += take;
		}
		return got;
	}
private:
	size_t nrec() const { return starts_.empty() ? 0 : starts_.size() - 1; }
	void unmap() { if(p_ && !inflated_.empty()) { inflated_.clear(); inflated_.shrink_to_fit(); } else if(p_) munmap((void*)p_, n_); p_ = nullptr; n_ = 0; starts_.clear(); cur_ = 0; }
	bool next_file() {
		unmap();
		if(fi_ >= files_.size()) return false;
		const std::string& fn = files_[fi_++];
		if(fn.size() > 3 && fn.compare(fn.size() - 3, 3, ".gz") == 0) {     // gzipped input (the reference reads it through zlib too)
			gzFile g = gzopen(fn.c_str(), "rb");
			if(!g) { fprintf(stderr, "Error: could not open %s\n", fn.c_str()); exit(1); }
			gzbuffer(g, 1 << 20);
			inflated_.clear();
			std::vector<char> chunk(8 << 20);
			int got;
			while((got = gzread(g, chunk.data(), (unsigned)chunk.size())) > 0) inflated_.insert(inflated_.end(), chunk.begin(), chunk.begin() + got);
			gzclose(g);
			if(inflated_.empty()) return true;
			p_ = inflated_.data(); n_ = inflated_.size();
		} else {
			const int fd = open(fn.c_str(), O_RDONLY);
			if(fd < 0) { fprintf(stderr, "Error: could not open %s\n", fn.c_str()); exit(1); }
			struct stat sb;
			fstat(fd, &sb);
			n_ = (size_t)sb.st_size;
			if(n_ == 0) { close(fd); return true; }
			p_ = (const char*)mmap(nullptr, n_, PROT_READ, MAP_PRIVATE, fd, 0);
			close(fd);
			if(p_ == MAP_FAILED) { fprintf(stderr, "Error: could not map %s\n", fn.c_str()); exit(1); }
		}
		const size_t T = std::min<size_t>((size_t)T_, n_ / (1 << 20) + 1);
		std::vector<std::vector<size_t> > loc(T);
		std::vector<size_t> nl(T + 1, 0);
		std::vector<std::thread> th;
		if(fasta_) {
			auto scan = [&](size_t t) {
				const size_t b = n_ * t / T, e = n_ * (t + 1) / T;
				for(size_t i = b; i < e; i++) if(p_[i] == '>' && (i == 0 || p_[i - 1] == '\n')) loc[t].push_back(i);
			};
			for(size_t t = 1; t < T; t++) th.emplace_back(scan, t);
			scan(0);
			for(auto& x : th) x.join();
			if(p_[0] != '>' && p_[0] != '#' && p_[0] != ';' && p_[0] != '\n' && p_[0] != '\r') { fprintf(stderr, "Error: reads file does not look like a FASTA file\n"); exit(1); }
		} else {
			auto cnt = [&](size_t t) { const size_t b = n_ * t / T, e = n_ * (t + 1) / T; size_t c = 0; for(size_t i = b; i < e; i++) c += p_[i] == '\n'; nl[t + 1] = c; };
			for(size_t t = 1; t < T; t++) th.emplace_back(cnt, t);
			cnt(0);
			for(auto& x : th) x.join();
			th.clear();
			for(size_t t = 0; t < T; t++) nl[t + 1] += nl[t];
			auto scan = [&](size_t t) {
				const size_t b = n_ * t / T, e = n_ * (t + 1) / T;
				size_t line = nl[t];                       // index of the line that starts after the next newline is line+1
				if(b == 0 && (line & 3) == 0) loc[t].push_back(0);
				for(size_t i = b; i < e; i++) if(p_[i] == '\n') { line++; if((line & 3) == 0 && i + 1 < n_) loc[t].push_back(i + 1); }
			};
			for(size_t t = 1; t < T; t++) th.emplace_back(scan, t);
			scan(0);
			for(auto& x : th) x.join();
			if(p_[0] != '@') { fprintf(stderr, "Error: reads file does not look like a FASTQ file\n"); exit(1); }
		}
		for(auto& v : loc) starts_.insert(starts_.end(), v.begin(), v.end());
		if(!fasta_) while(!starts_.empty() && (starts_.back() >= n_ || p_[starts_.back()] != '@')) starts_.pop_back();   // trailing blank lines
		starts_.push_back(n_);
		return true;
	}
	void parse_record(size_t r, Batch& b) const {
		const char* q = p_ + starts_[r];
		const char* end = p_ + starts_[r + 1];
		q++;                                                         // '>' or '@'
		const char* nm = q;
		while(q < end && *q != '\n') q++;
		size_t nlen = (size_t)(q - nm);
		if(nlen && nm[nlen - 1] == '\r') nlen--;
		if(nlen == 0) b.names += std::to_string(count_ + (r - cur_)); else b.names.append(nm, nlen);
		b.noffs.push_back((uint32_t)b.names.size());
		if(q < end) q++;
		const size_t c0 = b.codes.size();
		// -5 / -3 (gTrim5 / gTrim3, pat.cpp:820-832, 930-1010): bases dropped from the 5' / 3' end before alignment
		auto trim = [&]() {
			size_t L = b.codes.size() - c0;
			const size_t t5 = std::min<size_t>(trim5_, L);
			if(t5) { b.codes.erase(b.codes.begin() + c0, b.codes.begin() + c0 + t5); L -= t5; }
			const size_t t3 = std::min<size_t>(trim3_, L);
			if(t3) b.codes.resize(b.codes.size() - t3);
			return t5;
		};
		if(fasta_) {
			append_bases(b.codes, q, end, base_tables().fa);
			trim();
			b.offs.push_back((uint32_t)b.codes.size());
			return;
		}
		// FastqPatternSource::read (pat.cpp:932-945): '.' is N, every isalpha() character is a base through asc2dna
		// (alphabet.cpp:298: A C G T N, every other letter reads as A); anything else is skipped
		if(*(nm - 1) != '@') { fprintf(stderr, "Error: reads file does not look like a FASTQ file (record %llu does not start with '@'; wrapped records are not supported)\n", (unsigned long long)(count_ + (r - cur_))); exit(1); }
		{ const char* le = (const char*)memchr(q, '\n', (size_t)(end - q)); if(!le) le = end; append_bases(b.codes, q, le, base_tables().fq); q = le; }
		if(q + 1 < end && q[1] != '+') { fprintf(stderr, "Error: FASTQ record %.*s: the line after the sequence does not start with '+' (sequences wrapped over several lines are not supported)\n", (int)nlen, nm); exit(1); }
		const size_t Lraw = b.codes.size() - c0;
		const size_t t5 = trim();
		b.offs.push_back((uint32_t)b.codes.size());
		const size_t L = b.codes.size() - c0;
		if(q < end) q++;
		while(q < end && *q != '\n') q++;                            // '+' line
		if(q < end) q++;
		const char* ql = q;
		while(q < end && *q != '\n' && *q != '\r') q++;
		if((size_t)(q - ql) < Lraw) { fprintf(stderr, "Error: Read %.*s has more read characters than quality values.\n", (int)nlen, nm); exit(1); }
		b.quals.append(ql + t5, L);
		if(phred64_) {                                               // charToPhred33 qual.h:126-136
			for(size_t k = b.quals.size() - L; k < b.quals.size(); k++) {
				if(b.quals[k] < 64) { fprintf(stderr, "Saw ASCII character %d but expected 64-based Phred qual.\nTry not specifying --solexa1.3-quals/--phred64-quals.\n", (int)b.quals[k]); exit(1); }
				b.quals[k] = (char)(b.quals[k] - 31);
			}
		}
	}
public:
	bool phred64_ = false;
private:
	std::vector<std::string> files_;
	bool fasta_;
	int T_;
	uint32_t trim5_ = 0, trim3_ = 0;
	size_t fi_ = 0;
	const char* p_ = nullptr;
	size_t n_ = 0, cur_ = 0;
	std::vector<size_t> starts_;
	std::vector<char> inflated_;
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
	setenv("GPU_MAX_HW_QUEUES", "16", 0);   // before the HIP runtime initialises: the streams of one h2g stream run side by side (h2g_kernels.hip)
	std::string base, outfn, stats_fn;
	std::vector<std::string> u, m1, m2;
	bool fasta = false, nospliced = false, notempss = false, nohead = false, parse_only = false, no_unal = false;
	std::string known_ss, novel_ss, novel_out;
	bool tlen_adjust = true, use_haplotype = false;
	int max_alts_tried = 16, max_frag_len = 1000, min_frag_len = 0, pe_orientation = 0;
	bool nofw = false, norc = false, no_sq = false, omit_sec_seq = false;
	std::vector<std::pair<bool, std::string> > rg_args;
	bool new_summary = false;
	std::string summary_file;
	int chrname_mode = 0;
	bool phred64 = false, ignore_quals = false, quiet = false, raw_input = false, cmdline_input = false;
	bool report_mixed = true, report_discordant = true;
	bool dta = false, xs_only = false;
	int strandness = 0;
	uint64_t skip = 0, upto = ~0ull;
	uint32_t trim5 = 0, trim3 = 0;
	uint32_t dp = 0;
	size_t batch = 1u << 20;
	int device = 0, threads = 1, gpus = 1;
	uint32_t ss_window_opt = 0;
	std::string cmdline;
	std::vector<std::string> opts;                      // scoring / reporting options, applied once the index type is known
	bool sensitive = false, very_sensitive = false, saw_k = false;
	uint32_t k_arg = 0, max_seeds_arg = 0;
	for(int i = 0; i < argc; i++) { if(i) cmdline.push_back(' '); cmdline += argv[i]; }
	for(int i = 1; i < argc; i++) {
		const std::string a = argv[i];
		auto need = [&](const char* o) { if(i + 1 >= argc) { fprintf(stderr, "option %s needs an argument\n", o); exit(1); } return argv[++i]; };
		if(a == "-x") base = need("-x");
		else if(a == "-U") { auto v = split_commas(need("-U")); u.insert(u.end(), v.begin(), v.end()); }
		else if(a == "-1") { auto v = split_commas(need("-1")); m1.insert(m1.end(), v.begin(), v.end()); }
		else if(a == "-2") { auto v = split_commas(need("-2")); m2.insert(m2.end(), v.begin(), v.end()); }
		else if(a == "-S") outfn = need("-S");
		else if(a == "-r") raw_input = true;                                   // one sequence per line (RawPatternSource pat.h)
		else if(a == "-c") cmdline_input = true;                               // -U / -1 / -2 are comma-separated sequences (VectorPatternSource)
		else if(a == "-f") fasta = true;
		else if(a == "-q") fasta = false;
		else if(a == "-p" || a == "--threads") threads = atoi(need("-p"));        // host threads for parsing and SAM formatting
		else if(a == "--no-spliced-alignment") nospliced = true;
		else if(a == "--no-temp-splicesite") notempss = true;
		else if(a == "--ss-window") ss_window_opt = (uint32_t)strtoul(need("--ss-window"), nullptr, 10);   // reads a temporary splice site stays invisible for: 1000 x <-p> of the reference (hisat2.cpp:3687), decoupled from this program's host threads
		else if(a == "--dta" || a == "--downstream-transcriptome-assembly") dta = true;
		else if(a == "--dta-cufflinks") { dta = true; xs_only = true; }
		else if(a == "--rna-strandness") {
			const std::string v = need("--rna-strandness");
			strandness = v == "F" ? 1 : v == "R" ? 2 : v == "FR" ? 3 : v == "RF" ? 4 : 0;
			if(!strandness) { fprintf(stderr, "Error: should be one of F, R, FR, or RF \n"); return 1; }
		}
		else if(a == "--known-splicesite-infile") known_ss = need("--known-splicesite-infile");
		else if(a == "--novel-splicesite-infile") novel_ss = need("--novel-splicesite-infile");
		else if(a == "--novel-splicesite-outfile") novel_out = need("--novel-splicesite-outfile");
		else if(a == "--no-templatelen-adjustment") tlen_adjust = false;
		else if(a == "--max-altstried") { max_alts_tried = atoi(need("--max-altstried")); if(max_alts_tried < 8) { fprintf(stderr, "--max-altstried arg must be at least 8\n"); return 1; } }
		else if(a == "-X" || a == "--maxins") { max_frag_len = atoi(need("-X")); if(max_frag_len < 1) { fprintf(stderr, "-X arg must be at least 1\n"); return 1; } }
		else if(a == "-I" || a == "--minins") { min_frag_len = atoi(need("-I")); if(min_frag_len < 0) { fprintf(stderr, "-I arg must be positive\n"); return 1; } }
		else if(a == "--fr") pe_orientation = 0;                               // hisat2.cpp:1166-1168
		else if(a == "--rf") pe_orientation = 1;
		else if(a == "--ff") pe_orientation = 2;
		else if(a == "--nofw") nofw = true;                                    // hisat2.cpp:1337-1338
		else if(a == "--norc") norc = true;
		else if(a == "--rg-id") rg_args.push_back({true, need("--rg-id")});    // hisat2.cpp:1389-1407, in command-line order
		else if(a == "--rg") rg_args.push_back({false, need("--rg")});
		else if(a == "--no-sq" || a == "--sam-no-sq" || a == "--sam-nosq" || a == "--sam-noSQ") no_sq = true;
		else if(a == "--omit-sec-seq" || a == "--sam-omit-sec-seq") omit_sec_seq = true;
		else if(a == "--phred64" || a == "--phred64-quals" || a == "--solexa1.3-quals") phred64 = true;   // hisat2.cpp ARG_PHRED64
		else if(a == "--phred33" || a == "--phred33-quals") phred64 = false;
		else if(a == "--ignore-quals") ignore_quals = true;                    // hisat2.cpp:1434
		else if(a == "--remove-chrname") chrname_mode |= 1;
		else if(a == "--add-chrname") chrname_mode |= 2;
		else if(a == "--new-summary") new_summary = true;
		else if(a == "--summary-file") summary_file = need("--summary-file");
		else if(a == "--no-mixed") report_mixed = false;                       // hisat2.cpp:1162
		else if(a == "--no-discordant") report_discordant = false;             // hisat2.cpp:1161
		else if(a == "--haplotype") use_haplotype = true;                      // hisat2.cpp:1749 (ARG_HAPLOTYPE)
		else if(a == "--bowtie2-dp") dp = (uint32_t)atoi(need("--bowtie2-dp"));
		else if(a == "-k" || a == "--max-seeds" || a == "--mp" || a == "--sp" || a == "--np" || a == "--rdg" || a == "--rfg" || a == "--score-min" ||
		        a == "--min-intronlen" || a == "--max-intronlen" || a == "--pen-cansplice" || a == "--pen-noncansplice" ||
		        a == "--pen-canintronlen" || a == "--pen-intronlen" || a == "--pen-noncanintronlen") {
			opts.push_back(a); opts.push_back(need(a.c_str()));
		}
		else if(a == "--secondary" || a == "--no-softclip") opts.push_back(a);
		else if(a == "--sensitive") sensitive = true;
		else if(a == "--very-sensitive") very_sensitive = true;
		else if(a == "--no-hd" || a == "--no-head") nohead = true;
		else if(a == "--batch") batch = (size_t)atoll(need("--batch"));
		else if(a == "--device") device = atoi(need("--device"));
		else if(a == "--gpus") gpus = atoi(need("--gpus"));                        // batches round-robin over <int> devices, output in read order
		else if(a == "-s" || a == "--skip") skip = (uint64_t)atoll(need("-s"));     // skip the first <int> reads / pairs (hisat2.cpp:3319)
		else if(a == "-u" || a == "--upto" || a == "--qupto") upto = (uint64_t)atoll(need("-u"));
		else if(a == "-5" || a == "--trim5") trim5 = (uint32_t)atoi(need("-5"));
		else if(a == "-3" || a == "--trim3") trim3 = (uint32_t)atoi(need("-3"));
		else if(a == "--no-unal") no_unal = true;
		else if(a == "--quiet") quiet = true;                                      // gQuiet: no alignment summary on stderr (hisat2.cpp:4165)
		else if(a == "--version") { printf("hisat2-align-amd (h2g) — output format of HISAT2 2.2.3\n"); return 0; }
		else if(a == "--reorder" || a == "-t" || a == "--time" || a == "--mm" || a == "--qc-filter") {}   // output is always in read order; --mm (index mapping) and --qc-filter (a QSEQ field) have nothing to act on here
		else if(a == "--h2g-stats") stats_fn = need("--h2g-stats");               // writes {reads, second_pass, overflow} as JSON (tests, bench)
		else if(a == "--parse-only") parse_only = true;                           // test hook: ingest the reads, print counts + checksums
		else { fprintf(stderr, "hisat2-align-amd: option %s is not built (see DESIGN.md, scope)\n", a.c_str()); return 1; }
	}
	if(base.empty() || (u.empty() && (m1.empty() || m2.empty()))) {
		fprintf(stderr, "usage: hisat2-align-amd -x <ht2-base> {-U <r.fq> | -1 <m1.fq> -2 <m2.fq>} [-f|-q] --no-spliced-alignment [--bowtie2-dp 0|1|2] [-S out.sam]\n");
		return 1;
	}
	// -c / -r: the reads have no names (the reference numbers them, like FASTA records with an empty name) and no qualities ('I'): they are
	// handed to the FASTA reader as ">\n<sequence>\n" records through a temporary file
	static std::vector<std::string> tmp_inputs;   // (static: the exit handler below outlives main's frame)
	if(cmdline_input || raw_input) {
		auto as_fasta = [&](std::vector<std::string>& list) {
			if(list.empty()) return;
			std::string text;
			for(const std::string& item : list) {
				if(cmdline_input) { text += ">\n"; text += item; text += "\n"; continue; }
				FILE* f = fopen(item.c_str(), "rb");
				if(!f) { fprintf(stderr, "Error: could not open %s\n", item.c_str()); exit(1); }
				std::string line;
				int c;
				auto flush = [&]() { while(!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back(); if(!line.empty()) { text += ">\n"; text += line; text += "\n"; } line.clear(); };
				while((c = fgetc(f)) != EOF) { if(c == '\n') flush(); else line.push_back((char)c); }
				flush();
				fclose(f);
			}
			char path[] = "/tmp/h2g_reads_XXXXXX";
			const int fd = mkstemp(path);
			if(fd < 0 || write(fd, text.data(), text.size()) != (ssize_t)text.size()) { fprintf(stderr, "Error: cannot write a temporary read file\n"); exit(1); }
			close(fd);
			list.assign(1, path);
			tmp_inputs.push_back(path);
		};
		as_fasta(u); as_fasta(m1); as_fasta(m2);
		fasta = true;
	}
	// the -c / -r temporary read files go away on every way out, exit() included
	atexit([] { for(const std::string& p : tmp_inputs) unlink(p.c_str()); tmp_inputs.clear(); });
	struct TmpGuard { std::vector<std::string>& v; ~TmpGuard() { for(const std::string& p : v) unlink(p.c_str()); v.clear(); } } tmp_guard{tmp_inputs};
	if(parse_only) {
		Reader r(u.empty() ? m1 : u, fasta, threads);
		r.phred64_ = phred64;
		Batch b;
		uint64_t n = 0, bases = 0, h = 1469598103934665603ull;
		auto mix = [&](const void* p, size_t len) { const uint8_t* c = (const uint8_t*)p; for(size_t i = 0; i < len; i++) { h ^= c[i]; h *= 1099511628211ull; } };
		for(;;) {
			b.clear();
			const size_t got = r.fill(b, batch);
			if(!got) break;
			n += got; bases += b.codes.size();
			mix(b.codes.data(), b.codes.size()); mix(b.names.data(), b.names.size()); mix(b.quals.data(), b.quals.size());
			for(size_t i = 1; i <= got; i++) { const uint32_t l = b.offs[i] - b.offs[i - 1], nl = b.noffs[i] - b.noffs[i - 1]; mix(&l, 4); mix(&nl, 4); }
		}
		printf("%llu %llu %016llx\n", (unsigned long long)n, (unsigned long long)bases, (unsigned long long)h);
		return 0;
	}
	// Temporary splice sites (the reference's default): a read sees the junctions of reads at least W = 1000 * p ids before it
	// (hisat2.cpp:3687; -p 1 means W = 0, every read after the other).  The batches are waves of <= W reads run one after the other.
	const bool temp_ss = !nospliced && !notempss;
	uint32_t ss_window = 0, ss_wave = 0;      // the reference's visibility window, and the reads of one wave here (the window, or ONE read when it is 0)
	if(temp_ss) {
		// -p 1 (and no --ss-window): the reference's window is 0 (hisat2.cpp:3687) — a read sees the junctions of EVERY read before it, a strict
		// chain.  It runs as waves of one read: exact, and as slow as a chain is (a device round trip per read); meant for small inputs — the
		// reference's bare default invocation `hisat2 -x idx -U reads` then simply works.  -p >= 2 / --ss-window are the throughput modes.
		ss_window = ss_window_opt ? ss_window_opt : (threads < 2 ? 0u : 1000u * (uint32_t)threads);
		ss_wave = ss_window ? ss_window : 1u;
		// a wave sizes the streams and the result rows (one shard per device): bound it — the reference's own window at its largest useful
		// -p (1000 x 256 threads) is far below this, and an absurd value would only be an allocation failure later
		const uint32_t ss_wave_max = 4u * 1024u * 1024u;
		if(ss_wave > ss_wave_max) {
			fprintf(stderr, "hisat2-align-amd: --ss-window %u (or 1000 x -p) makes waves of more than %u reads; a wave is one resident batch per device "
			                "(streams and result rows are sized by it, --batch does not apply to the temporary-splice-site mode): use --ss-window <= %u, "
			                "or --no-temp-splicesite with --batch\n", ss_wave, ss_wave_max, ss_wave_max);
			return 1;
		}
		batch = ss_wave;   // a wave is exactly one shard per device (want()): a smaller --batch would complete shards (and merge their junctions) in the middle of a wave
	}
	const bool paired = u.empty();
	const double t0 = now();
	h2g_load_opts lo; h2g_load_opts_init(&lo); lo.device = device; lo.load_local = 1;
	// --gpus N: one index replica and one stream per device; batch k runs on device k mod N while the others are in flight, and
	// the batches are completed (fetched, formatted, written) strictly in order, so the output is the single-GPU output.
	// H2G_GPUS_SHARE_DEVICE=1 (test hook) lets the N streams share the devices that exist.
	if(gpus < 1) gpus = 1;
	int ndev = h2g_device_count();
	if(ndev < 1) die("no GPU");
	if(gpus > ndev && !getenv("H2G_GPUS_SHARE_DEVICE")) { fprintf(stderr, "hisat2-align-amd: --gpus %d but %d device(s) visible\n", gpus, ndev); return 1; }
	// One stream per device.  (H2G_STREAMS_PER_DEVICE=2 puts batch k - 1 on the device while the main thread fetches and formats batch k - 2: measured on 10 M pairs, E. coli-size
	// index — 2.77 s against 2.74 s to /dev/null, and slower to a file: the kernels are 0.16 s of the run, there is nothing to hide; profiles/r05_NOTES.md §12.)
	const int ndevs_asked = gpus;
	{
		const char* e = getenv("H2G_STREAMS_PER_DEVICE");
		const int per = e && !temp_ss ? atoi(e) : 1;
		if(per > 1) gpus *= per;
	}
	std::vector<h2g_index*> ixs((size_t)gpus, nullptr);
	for(int g = 0; g < gpus; g++) {
		const int dev = (device + g % ndevs_asked) % ndev;
		for(int q = 0; q < g; q++) if((device + q % ndevs_asked) % ndev == dev) ixs[g] = ixs[q];      // shared device: share the replica
		if(ixs[g]) continue;
		lo.device = dev;
		if(h2g_index_load(base.c_str(), &lo, &ixs[g]) != H2G_OK) die("cannot load the index onto the GPU");
	}
	h2g_index* ix = ixs[0];
	h2g_sam* sam = nullptr;
	if(h2g_sam_open(base.c_str(), &sam) != H2G_OK) die("cannot read reference names");
	if(chrname_mode == 3) { fprintf(stderr, "Error: --remove-chrname and --add-chrname cannot be used at the same time\n"); return 1; }   // hisat2.cpp:3958
	if(chrname_mode) h2g_sam_set_chrname_mode(sam, chrname_mode);
	h2g_align_params P; h2g_align_params_init(&P, ix);
	P.bowtie2_dp = dp;
	P.no_spliced_alignment = nospliced ? 1 : 0; P.no_temp_splicesite = notempss ? 1 : 0;
	for(size_t i = 0; i < opts.size(); i++) {             // same parse rules as hisat2.cpp:1500-1620 / aligner_seed_policy.cpp
		const std::string& o = opts[i];
		auto two = [&](int32_t* x, int32_t* y) { const std::string& v = opts[++i]; *x = atoi(v.c_str()); const size_t c = v.find(','); if(c != std::string::npos) *y = atoi(v.c_str() + c + 1); };
		if(o == "-k") {
			const int k = atoi(opts[++i].c_str());
			if(k < 1) { fprintf(stderr, "-k arg must be at least 1\n"); return 1; }
			k_arg = (uint32_t)k; saw_k = true;
		}
		else if(o == "--max-seeds") max_seeds_arg = (uint32_t)atoi(opts[++i].c_str());
		else if(o == "--secondary") P.secondary = 1;
		else if(o == "--mp") two(&P.mm_max, &P.mm_min);
		else if(o == "--sp") { int32_t unused = 0; two(&P.sc_max, &unused); P.sc_min = P.sc_max; }   // both read from the first number (aligner_seed_policy.cpp:438)
		else if(o == "--no-softclip") P.sc_max = P.sc_min = INT32_MAX;
		else if(o == "--np") P.n_pen = atoi(opts[++i].c_str());
		else if(o == "--rdg") two(&P.rdg_const, &P.rdg_linear);
		else if(o == "--rfg") two(&P.rfg_const, &P.rfg_linear);
		else if(o == "--score-min") {
			const std::string& v = opts[++i];
			P.score_min_type = v[0] == 'C' ? 1 : v[0] == 'L' ? 2 : v[0] == 'S' ? 3 : v[0] == 'G' ? 4 : 0;
			if(!P.score_min_type) { fprintf(stderr, "Error: bad function type in --score-min %s\n", v.c_str()); return 1; }
			P.score_min_const = P.score_min_coeff = 0.0;
			const size_t c1 = v.find(',');
			if(c1 != std::string::npos) { P.score_min_const = atof(v.c_str() + c1 + 1); const size_t c2 = v.find(',', c1 + 1); if(c2 != std::string::npos) P.score_min_coeff = atof(v.c_str() + c2 + 1); }
		}
		// splice scoring hisat2.cpp:1631-1688
		else if(o == "--min-intronlen" || o == "--max-intronlen") {
			const int v = atoi(opts[++i].c_str());
			if(v < 20) { fprintf(stderr, "%s arg must be at least 20\n", o.c_str()); return 1; }
			(o == "--min-intronlen" ? P.min_intronlen : P.max_intronlen) = (uint32_t)v;
		}
		else if(o == "--pen-cansplice" || o == "--pen-noncansplice") {
			const int v = atoi(opts[++i].c_str());
			if(v < 0) { fprintf(stderr, "%s arg must be at least 0\n", o.c_str()); return 1; }
			(o == "--pen-cansplice" ? P.pen_cansplice : P.pen_noncansplice) = v;
		}
		else if(o == "--pen-canintronlen" || o == "--pen-intronlen" || o == "--pen-noncanintronlen") {   // PARSE_FUNC: only the given fields change
			const bool nc = o == "--pen-noncanintronlen";
			const std::string& v = opts[++i];
			const uint32_t t = v[0] == 'C' ? 1 : v[0] == 'L' ? 2 : v[0] == 'S' ? 3 : v[0] == 'G' ? 4 : 0;
			if(!t) { fprintf(stderr, "Error: bad function type in %s %s\n", o.c_str(), v.c_str()); return 1; }
			(nc ? P.pen_noncanintronlen_type : P.pen_canintronlen_type) = t;
			const size_t c1 = v.find(',');
			if(c1 != std::string::npos) {
				(nc ? P.pen_noncanintronlen_const : P.pen_canintronlen_const) = atof(v.c_str() + c1 + 1);
				const size_t c2 = v.find(',', c1 + 1);
				if(c2 != std::string::npos) (nc ? P.pen_noncanintronlen_coeff : P.pen_canintronlen_coeff) = atof(v.c_str() + c2 + 1);
			}
		}
	}
	// presets and the -k / --max-seeds defaults are resolved after every option was read, whatever their order (hisat2.cpp:1882-1909, 3903)
	if(dta) {   // hisat2.cpp:3920, 4078-4079: after every option was read
		P.min_anchor_len = 15; P.min_anchor_len_noncan = 20;
		P.pen_noncanintronlen_type = 4; P.pen_noncanintronlen_const = -8.0; P.pen_noncanintronlen_coeff = 2.0;
	}
	{   // COST_MODEL_CONSTANT: every mismatch costs the maximum (aligner_seed_policy.cpp:279, scoring.h:129); a --mp sets the quality model again (:418)
		bool saw_mp = false;
		for(const std::string& o : opts) if(o == "--mp") saw_mp = true;
		if(ignore_quals && !saw_mp) P.mm_min = P.mm_max;
	}
	P.xs_only = xs_only ? 1 : 0;
	P.use_haplotype = use_haplotype ? 1 : 0;
	P.max_alts_tried = (uint32_t)max_alts_tried;
	P.max_frag_len = (uint32_t)max_frag_len;
	P.min_frag_len = (uint32_t)min_frag_len; P.pe_orientation = (uint32_t)pe_orientation; P.nofw = nofw ? 1 : 0; P.norc = norc ? 1 : 0;
	h2g_align_params_presets(&P, ix, saw_k ? 1 : 0, k_arg, max_seeds_arg, sensitive ? 1 : 0, very_sensitive ? 1 : 0);
	if(!P.no_spliced_alignment && P.max_intronlen > 0xfffffu) {
		fprintf(stderr, "hisat2-align-amd: --max-intronlen %u is beyond the 1048575 bases a splice edit holds here\n", P.max_intronlen);
		return 1;
	}
	if(P.min_intronlen > P.max_intronlen) {   // hisat2.cpp:4278
		fprintf(stderr, "--min-intronlen(%u) should not be greater than --max-intronlen(%u)\n", P.min_intronlen, P.max_intronlen);
		return 1;
	}
	if(P.khits > 30 || P.kseeds > 64 || P.kseeds < P.khits) {
		fprintf(stderr, "hisat2-align-amd: -k %u / --max-seeds %u is outside the built range (-k <= 30, -k <= --max-seeds <= 64)\n", P.khits, P.kseeds);
		return 1;
	}
	// splice sites from files (hisat2.cpp:4100-4120): one database for go() on every device and for TLEN
	std::vector<h2g_splice_site> sites;                  // the splice-site database: file sites, then the temporary ones by first appearance
	std::map<std::array<uint32_t, 4>, size_t> site_at;    // (text, left, right, dir) -> position in `sites`
	auto publish_sites = [&]() {
		for(int g = 0; g < gpus; g++) {
			bool first = true;
			for(int q = 0; q < g; q++) if(ixs[(size_t)q] == ixs[(size_t)g]) first = false;
			if(first && h2g_index_set_splice_sites(ixs[(size_t)g], sites.data(), sites.size(), ss_window) != H2G_OK) die("cannot upload the splice sites");
		}
		h2g_sam_set_splice_sites(sam, sites.data(), sites.size(), ss_window);
	};
	if(!nospliced && (!known_ss.empty() || !novel_ss.empty())) {
		for(int pass = 0; pass < 2; pass++) {
			const std::string& fn = pass == 0 ? known_ss : novel_ss;
			if(fn.empty()) continue;
			const size_t n = h2g_sam_read_splice_site_file(sam, fn.c_str(), pass == 0, nullptr, 0);
			if(n == (size_t)-1) { fprintf(stderr, "Error: Could not open %s\n", fn.c_str()); return 1; }
			const size_t at = sites.size();
			sites.resize(at + n);
			h2g_sam_read_splice_site_file(sam, fn.c_str(), pass == 0, sites.data() + at, n);
		}
		{   // SpliceSiteDB::read keeps the first of equal sites (splice_site.cpp:750)
			std::vector<h2g_splice_site> uniq;
			for(const h2g_splice_site& x : sites) {
				const std::array<uint32_t, 4> key = {x.tidx, x.left, x.right, (uint32_t)x.dir};
				if(site_at.emplace(key, uniq.size()).second) uniq.push_back(x);
			}
			sites.swap(uniq);
		}
		publish_sites();
	} else if(temp_ss) publish_sites();                    // (the window of the wave scheme; the sites arrive wave after wave)
	if(!temp_ss && !nospliced && !novel_out.empty() && h2g_sam_novel_splice_sites_text(sam, nullptr, 0) > 0) {
		// write (the outfile) + read (a file's or the index's sites) without the temporary-site window: the reference then lets every read see
		// the junctions of whichever reads its threads happened to finish first (window 0, hisat2.cpp:3687, :4092-4093) — not a function of the input
		fprintf(stderr, "hisat2-align-amd: --novel-splicesite-outfile with --no-temp-splicesite and a splice-site database (file or --ss index) "
		        "makes the reference's output depend on thread timing; drop --no-temp-splicesite (output == hisat2 -p <int> --reorder)\n");
		return 1;
	}
	if(temp_ss || (!nospliced && !novel_out.empty())) h2g_sam_collect_novel_sites(sam, 1);   // SpliceSiteDB's `write` (hisat2.cpp:4092)
	h2g_sam_set_templatelen_adjustment(sam, tlen_adjust);
	h2g_sam_set_report_policy(sam, report_discordant, report_mixed);
	for(const auto& r : rg_args) h2g_sam_add_read_group(sam, r.first ? r.second.c_str() : nullptr, r.first ? nullptr : r.second.c_str());
	h2g_sam_set_header_options(sam, no_sq, omit_sec_seq);
	h2g_sam_set_new_summary(sam, new_summary);
	h2g_sam_set_score_min(sam, P.score_min_type, P.score_min_const, P.score_min_coeff);
	h2g_sam_set_secondary(sam, (int)P.secondary);
	h2g_sam_set_rna_strandness(sam, strandness);
	FILE* out = outfn.empty() ? stdout : fopen(outfn.c_str(), "wb");
	if(!out) { fprintf(stderr, "cannot open %s\n", outfn.c_str()); return 1; }
	// output text buffer: raw storage, grown without value-initialising hundreds of MB per batch
	struct RawBuf { char* p = nullptr; size_t n = 0; void resize(size_t m) { if(m > n) { free(p); p = (char*)malloc(m); n = m; if(!p) { fprintf(stderr, "out of memory\n"); exit(1); } } } char* data() { return p; } size_t size() const { return n; } ~RawBuf() { free(p); } };
	RawBuf hdr_buf;
	RawBuf& buf = hdr_buf;          // (the header; the batches' text goes through the writer's ring below)
	buf.resize(1 << 20);
	if(!nohead) {
		const size_t need = h2g_sam_header(sam, cmdline.c_str(), nullptr, 0);
		buf.resize(need + 1);
		h2g_sam_header(sam, cmdline.c_str(), buf.data(), buf.size());
		fwrite(buf.data(), 1, need, out);
	}
	const double t1 = now();
	h2g_sam_set_threads(sam, threads);
	h2g_sam_set_no_unal(sam, no_unal ? 1 : 0);
	Reader ra(paired ? m1 : u, fasta, threads, trim5, trim3), rb(m2, fasta, threads, trim5, trim3);
	ra.phred64_ = rb.phred64_ = phred64;
	if(skip) {   // -s: the skipped reads are parsed (their ids count) but not aligned
		Batch junk;
		for(uint64_t left = skip; left > 0;) { junk.clear(); const size_t g = ra.fill(junk, (size_t)std::min<uint64_t>(left, batch)); if(paired) { junk.clear(); rb.fill(junk, g); } if(!g) break; left -= g; }
	}
	uint64_t budget = upto;                               // -u counts the reads after the skipped ones (qUpto += skipReads, hisat2.cpp:1959-1963)
	// Formatting on a thread of its own (round 6): the main thread fetches batch k + 1's records while batch k's text is written — what the device returns goes to one of two sets of page-locked
	// buffers, the formatter works through them in order.  Not with temporary splice sites / a novel-site file: there a batch's junctions must be in the database before the next wave starts.
	const bool async_fmt = !temp_ss && novel_out.empty() && !(getenv("H2G_CLI_ASYNC_FMT") && atoi(getenv("H2G_CLI_ASYNC_FMT")) == 0);
	const int G = gpus, H = gpus + (async_fmt ? 3 : 2);  // G streams (one per device) in flight, H host batch buffers: batch k + 1 is parsed
	std::vector<Batch> A((size_t)H), B((size_t)H);     // (on a thread of its own) into buffer (k + 1) mod H while batch k is uploaded and up to G earlier ones are on the GPUs / being written
	struct Str { h2g_stream* st = nullptr; size_t reads = 0, bases = 0; long batch = -1; size_t n = 0; uint64_t first_id = 0; };
	uint64_t next_id = skip;                              // Read::rdid of the next read (the skipped ones count, hisat2.cpp:3319)
	std::vector<Str> S((size_t)G);
	uint64_t nreads = 0, naligned = 0, novf = 0, nsecond = 0;
	double t_gpu = 0, t_fmt = 0, t_parse = 0, t_up = 0, t_fetch = 0, t_stream = 0;
	// what comes back from the device lands in page-locked memory (h2g_host_alloc): the copies run at the link's rate.  The records travel compact
	// (h2g_align_*_fetch_compact: 40 bytes + 12 per edit held instead of 424 per record) and are formatted in that layout.
	struct Pinned {
		uint8_t* p = nullptr; size_t cap = 0;
		void need(size_t n) { if(n <= cap) return; h2g_host_free(p); cap = n + n / 4 + 4096; p = (uint8_t*)h2g_host_alloc(cap); if(!p) { fprintf(stderr, "hisat2-align-amd: cannot allocate %zu bytes of page-locked memory\n", cap); exit(1); } }
		~Pinned() { h2g_host_free(p); }
	};
	struct PinSet { Pinned res, rec1, rec2, o1, o2; std::vector<h2g_edit> long_edits; size_t nlong = 0; };
	PinSet pins[2];
	std::string ovf_names;
	// Temporary splice sites on G devices: a wave of W reads is cut into G shards that run side by side — a read never sees the junctions of
	// its own wave (readid + W > its id), so the shards need nothing from one another; every shard's junctions join the database (on every
	// device) before the next wave starts (the exchange between two waves is the junction list, tens of bytes per site).
	size_t wave_left = ss_wave;                           // reads the current wave still takes
	// ---- the writer: the text of a batch goes to the output on a thread of its own (6 GB of SAM per 10 M pairs: a third of the run when the main thread wrote it).
	// Three text buffers go round; the batches are written in the order they were formatted (one writer, a FIFO).
	RawBuf wtext[3];
	size_t wused[3] = {0, 0, 0};
	std::mutex wm; std::condition_variable wcv;
	std::deque<int> wqueue, wfree = {0, 1, 2};
	bool wdone = false, werr = false;
	std::thread writer([&]() {
		for(;;) {
			int i;
			{ std::unique_lock<std::mutex> lk(wm); wcv.wait(lk, [&] { return !wqueue.empty() || wdone; }); if(wqueue.empty()) return; i = wqueue.front(); wqueue.pop_front(); }
			if(wused[i] && fwrite(wtext[i].data(), 1, wused[i], out) != wused[i]) werr = true;
			{ std::lock_guard<std::mutex> lk(wm); wfree.push_back(i); }
			wcv.notify_all();
		}
	});
	auto wacquire = [&]() { std::unique_lock<std::mutex> lk(wm); wcv.wait(lk, [&] { return !wfree.empty(); }); const int i = wfree.front(); wfree.pop_front(); return i; };
	auto wsubmit = [&](int i, size_t used_) { { std::lock_guard<std::mutex> lk(wm); wused[i] = used_; wqueue.push_back(i); } wcv.notify_all(); };
	auto wfinish = [&]() { { std::lock_guard<std::mutex> lk(wm); wdone = true; } wcv.notify_all(); writer.join(); };
	// ---- the parser: batch j is read into buffer j mod H as soon as that buffer is free (batch j - H is written), ahead of the main thread
	std::mutex pm; std::condition_variable pcv;
	long parsed = 0, completed_cnt = 0;
	std::vector<size_t> pn((size_t)H, 0);
	bool perr = false;
	double t_parse_busy = 0;
	// the formatter's queue: jobs in fetch order; pinned set j is free again once its job has been formatted
	struct FmtJob { long batch; size_t n; uint64_t first_id; int set; };
	std::mutex fm; std::condition_variable fcv;
	std::deque<FmtJob> fqueue;
	bool set_busy[2] = {false, false}, fdone = false;
	long nfetched = 0;
	// format + hand to the writer: the batch whose records lie in pinned set `job.set`
	auto format_job = [&](const FmtJob& job) {
		Batch& a = A[(size_t)(job.batch % H)]; Batch& b = B[(size_t)(job.batch % H)];
		PinSet& ps = pins[job.set];
		const size_t n = job.n;
		size_t used = 0;
		const int wi = wacquire();
		RawBuf& buf = wtext[wi];
		const double tf = now();
		h2g_sam_set_first_read_id(sam, job.first_id);
		h2g_sam_set_long_edits(sam, ps.nlong ? ps.long_edits.data() : nullptr, ps.nlong);
		if(paired) {
			h2g_pair_result* pres = (h2g_pair_result*)ps.res.p;
			uint64_t *ao1 = (uint64_t*)ps.o1.p, *ao2 = (uint64_t*)ps.o2.p;
			buf.resize(n * 1400 + 6 * (a.codes.size() + b.codes.size()) + 4096);
			h2g_status rc = h2g_sam_format_paired_compact(sam, a.codes.data(), a.offs.data(), a.have_quals ? a.quals.data() : nullptr, a.names.data(), a.noffs.data(),
			                                      b.codes.data(), b.offs.data(), b.have_quals ? b.quals.data() : nullptr, b.names.data(), b.noffs.data(), n,
			                                      pres, ps.rec1.p, ao1, ps.rec2.p, ao2, P.khits, buf.data(), buf.size(), &used);
			if(rc != H2G_OK) {
				buf.resize(used + 16);
				rc = h2g_sam_format_paired_compact(sam, a.codes.data(), a.offs.data(), a.have_quals ? a.quals.data() : nullptr, a.names.data(), a.noffs.data(),
				                           b.codes.data(), b.offs.data(), b.have_quals ? b.quals.data() : nullptr, b.names.data(), b.noffs.data(), n,
				                           pres, ps.rec1.p, ao1, ps.rec2.p, ao2, P.khits, buf.data(), buf.size(), &used);
				if(rc != H2G_OK) die("h2g_sam_format_paired_compact");
			}
			for(size_t i = 0; i < n; i++) { naligned += pres[i].npairs > 0; if(pres[i].overflow) { novf++; if(ovf_names.size() < 4096) { ovf_names.append(a.names.data() + a.noffs[i], a.noffs[i + 1] - a.noffs[i]); ovf_names += " (bits " + std::to_string(pres[i].overflow) + ")\n"; } } }
		} else {
			h2g_read_result* res = (h2g_read_result*)ps.res.p;
			uint64_t* ao1 = (uint64_t*)ps.o1.p;
			buf.resize(n * 700 + 3 * a.codes.size() + 4096);
			h2g_status rc = h2g_sam_format_unpaired_compact(sam, a.codes.data(), a.offs.data(), a.have_quals ? a.quals.data() : nullptr, a.names.data(), a.noffs.data(), n,
			                                        res, ps.rec1.p, ao1, buf.data(), buf.size(), &used);
			if(rc != H2G_OK) {
				buf.resize(used + 16);
				rc = h2g_sam_format_unpaired_compact(sam, a.codes.data(), a.offs.data(), a.have_quals ? a.quals.data() : nullptr, a.names.data(), a.noffs.data(), n,
				                             res, ps.rec1.p, ao1, buf.data(), buf.size(), &used);
				if(rc != H2G_OK) die("h2g_sam_format_unpaired_compact");
			}
			for(size_t i = 0; i < n; i++) { naligned += res[i].nselect > 0; if(res[i].overflow) { novf++; if(ovf_names.size() < 4096) { ovf_names.append(a.names.data() + a.noffs[i], a.noffs[i + 1] - a.noffs[i]); ovf_names += " (bits " + std::to_string(res[i].overflow) + ")\n"; } } }
		}
		t_fmt += now() - tf;
		wsubmit(wi, used);
		if(temp_ss || !novel_out.empty()) {   // the junctions of the lines just written join the database (SpliceSiteDB::addSpliceSite: smallest read id per site)
			static std::vector<h2g_splice_site> novel;
			const size_t k = h2g_sam_take_novel_sites(sam, nullptr, 0);
			novel.resize(k);
			if(k) h2g_sam_take_novel_sites(sam, novel.data(), k);
			// only what is new (or whose smallest read id went down) goes to the devices and the formatter: they merge it into their sorted
			// copies (h2g_index_add_splice_sites) — the cost of a wave is its own junctions, not the database's size
			static std::vector<h2g_splice_site> delta;
			delta.clear();
			if(temp_ss) for(const h2g_splice_site& x : novel) {
				const std::array<uint32_t, 4> key = {x.tidx, x.left, x.right, (uint32_t)x.dir};
				auto it = site_at.find(key);
				if(it == site_at.end()) { site_at.emplace(key, sites.size()); sites.push_back(x); delta.push_back(x); }
				else if(!sites[it->second].fromfile && x.readid < sites[it->second].readid) { sites[it->second].readid = x.readid; delta.push_back(sites[it->second]); }
			}
			if(!delta.empty()) {
				for(int g2 = 0; g2 < gpus; g2++) {
					bool first = true;
					for(int q = 0; q < g2; q++) if(ixs[(size_t)q] == ixs[(size_t)g2]) first = false;
					if(first && h2g_index_add_splice_sites(ixs[(size_t)g2], delta.data(), delta.size()) != H2G_OK) die("cannot upload the splice sites");
				}
				h2g_sam_add_splice_sites(sam, delta.data(), delta.size());
			}
		}
		nreads += n;
		{ std::lock_guard<std::mutex> lk(pm); completed_cnt++; }      // (its read buffers are free for the parser)
		pcv.notify_all();
	};
	std::thread formatter;
	if(async_fmt) formatter = std::thread([&]() {
		for(;;) {
			FmtJob job;
			{ std::unique_lock<std::mutex> lk(fm); fcv.wait(lk, [&] { return !fqueue.empty() || fdone; }); if(fqueue.empty()) return; job = fqueue.front(); fqueue.pop_front(); }
			format_job(job);
			{ std::lock_guard<std::mutex> lk(fm); set_busy[job.set] = false; }
			fcv.notify_all();
		}
	});
	auto ffinish = [&]() { if(formatter.joinable()) { { std::lock_guard<std::mutex> lk(fm); fdone = true; } fcv.notify_all(); formatter.join(); } };
	// fetch (+ format + write, or hand to the formatter) the batch that stream `g` carries
	auto complete = [&](int g) {
		Str& sg = S[(size_t)g];
		if(sg.batch < 0) return;
		h2g_stream* st = sg.st;
		const size_t n = sg.n;
		const int set = (int)(nfetched % 2);
		if(async_fmt) { std::unique_lock<std::mutex> lk(fm); fcv.wait(lk, [&] { return !set_busy[set]; }); set_busy[set] = true; }
		PinSet& ps = pins[set];
		const double tq0 = now();
		{	// records with more than H2G_MAX_EDITS edits (long deletions: one edit per base) keep their lists in the stream's long-edit area
			size_t nl = 0;
			h2g_status lrc = h2g_align_fetch_long_edits(st, nullptr, 0, &nl);
			if(nl) { ps.long_edits.resize(nl); lrc = h2g_align_fetch_long_edits(st, ps.long_edits.data(), ps.long_edits.size(), &nl); }
			if(lrc != H2G_OK) die("h2g_align_fetch_long_edits");
			ps.nlong = nl;
		}
		if(paired) {
			ps.res.need(n * sizeof(h2g_pair_result)); ps.o1.need((n + 1) * 8); ps.o2.need((n + 1) * 8);
			ps.rec1.need(n * 64 + 4096); ps.rec2.need(n * 64 + 4096);
			h2g_pair_result* pres = (h2g_pair_result*)ps.res.p;
			uint64_t *ao1 = (uint64_t*)ps.o1.p, *ao2 = (uint64_t*)ps.o2.p;
			ao1[n] = 0; ao2[n] = 0;
			if(const h2g_status frc = h2g_align_pairs_fetch_compact(st, pres, ps.rec1.p, ps.rec1.cap, ao1, ps.rec2.p, ps.rec2.cap, ao2, 0, n); frc != H2G_OK) {
				// one retry, and only for "buffer too small": H2G_ERR_ARG with the bytes needed in boffs[n] (zeroed above: page-locked memory starts uninitialised)
				if(frc != H2G_ERR_ARG || (ao1[n] <= ps.rec1.cap && ao2[n] <= ps.rec2.cap)) die("h2g_align_pairs_fetch_compact");
				ps.rec1.need(ao1[n] + 8); ps.rec2.need(ao2[n] + 8);
				if(h2g_align_pairs_fetch_compact(st, pres, ps.rec1.p, ps.rec1.cap, ao1, ps.rec2.p, ps.rec2.cap, ao2, 0, n) != H2G_OK) die("h2g_align_pairs_fetch_compact");
			}
		} else {
			ps.res.need(n * sizeof(h2g_read_result)); ps.o1.need((n + 1) * 8); ps.rec1.need(n * 64 + 4096);
			h2g_read_result* res = (h2g_read_result*)ps.res.p;
			uint64_t* ao1 = (uint64_t*)ps.o1.p;
			ao1[n] = 0;
			if(const h2g_status frc = h2g_align_fetch_compact(st, res, ps.rec1.p, ps.rec1.cap, ao1, 0, n); frc != H2G_OK) {
				if(frc != H2G_ERR_ARG || ao1[n] <= ps.rec1.cap) die("h2g_align_fetch_compact");
				ps.rec1.need(ao1[n] + 8);
				if(h2g_align_fetch_compact(st, res, ps.rec1.p, ps.rec1.cap, ao1, 0, n) != H2G_OK) die("h2g_align_fetch_compact");
			}
		}
		{ h2g_counters hc; if(h2g_get_counters(st, &hc) == H2G_OK) nsecond += hc.n_second_pass; }
		t_fetch += now() - tq0;
		const FmtJob job{sg.batch, n, sg.first_id, set};
		nfetched++;
		sg.batch = -1;                                  // (the stream's rows are copied: it can take the next batch)
		if(async_fmt) { { std::lock_guard<std::mutex> lk(fm); fqueue.push_back(job); } fcv.notify_all(); }
		else format_job(job);
	};
	std::thread parser([&]() {
		uint64_t pbudget = budget;
		size_t pwave_left = ss_wave;
		for(long j = 0;; j++) {
			{ std::unique_lock<std::mutex> lk(pm); pcv.wait(lk, [&] { return j < completed_cnt + H; }); }
			Batch& a = A[(size_t)(j % H)]; Batch& b = B[(size_t)(j % H)];
			a.clear(); b.clear();
			const double tp = now();
			size_t w = (size_t)std::min<uint64_t>(batch, pbudget);
			if(temp_ss) { const size_t shard = (ss_wave + (size_t)gpus - 1) / (size_t)gpus; w = std::min(w, std::min(shard, pwave_left)); }
			// the two mate files are parsed side by side (each fill is threaded in itself; one after the other they were a second per 10 M pairs, and the main thread waited for them)
			size_t nb_ = 0;
			std::thread tb;
			const size_t wb_ = pbudget ? w : 0;
			if(paired && wb_) tb = std::thread([&]() { nb_ = rb.fill(b, wb_); });
			const size_t n = pbudget ? ra.fill(a, w) : 0;
			if(tb.joinable()) tb.join();
			pbudget -= std::min<uint64_t>(pbudget, n);
			const bool bad = paired && nb_ < n;                      // (-2 ran out before -1: the reference's error; a longer -2 is not looked at, as before)
			if(temp_ss) { pwave_left -= n; if(pwave_left == 0) pwave_left = ss_wave; }
			t_parse_busy += now() - tp;
			{ std::lock_guard<std::mutex> lk(pm); pn[(size_t)(j % H)] = n; perr = perr || bad; parsed = j + 1; }
			pcv.notify_all();
			if(n == 0 || bad) return;
		}
	});
	struct Joiner { std::thread& t; ~Joiner() { if(t.joinable()) t.detach(); } } pjoin{parser}, wjoin{writer}, fjoin{formatter};      // (an early `return` / exit leaves no joinable thread behind)
	for(long k = 0;; k++) {
		Batch& a = A[(size_t)(k % H)]; Batch& b = B[(size_t)(k % H)];
		double tp = now();
		size_t n;
		bool perr_now;
		{ std::unique_lock<std::mutex> lk(pm); pcv.wait(lk, [&] { return parsed > k; }); n = pn[(size_t)(k % H)]; perr_now = perr; }
		if(perr_now) {
			// the parser has returned (it stops at the short file); the writer waits on a condition variable that lives in this frame: both threads are
			// joined before the frame goes (a detached waiter would block the variable's destructor for ever)
			fprintf(stderr, "Error, fewer reads in file specified with -2 than in file specified with -1\n");
			parser.join();
			ffinish();
			wfinish();
			return 1;
		}
		t_parse += now() - tp;                         // (what the main thread waited for the parser)
		if(n == 0) break;
		const int g = (int)(k % G);
		const double tg = now();
		if(temp_ss) {                                  // a wave needs the sites of every earlier one: nothing of them stays in flight when it starts
			if(wave_left == ss_wave) for(long q = k - G; q < k; q++) if(q >= 0) complete((int)(q % G));
			wave_left -= n;
			if(wave_left == 0) wave_left = ss_wave;
		}
		complete(g);                                   // the batch this stream still carries (k - G): the oldest one in flight
		Str& sg = S[(size_t)g];
		size_t bases = a.codes.size();
		if(paired && b.codes.size() > bases) bases = b.codes.size();
		if(!sg.st || n > sg.reads || bases > sg.bases) {
			if(sg.st) h2g_stream_free(sg.st);
			sg.reads = n > batch ? n : batch; sg.bases = bases + bases / 4 + 1024;
			const double ts = now();
			if(h2g_stream_create(ixs[(size_t)g], sg.reads, sg.bases, &sg.st) != H2G_OK) die("cannot create the device stream");
			t_stream += now() - ts;
		}
		const double tq0 = now();
		if(h2g_set_reads(sg.st, a.codes.data(), a.offs.data(), a.have_quals ? a.quals.data() : nullptr, n) != H2G_OK) die("h2g_set_reads");
		if(h2g_set_read_names(sg.st, a.names.data(), a.noffs.data(), n) != H2G_OK) die("h2g_set_read_names");
		// read ids are 32 bits in the splice-site window test (DSpliceSite::readid): past that the temporary sites' visibility would wrap silently
		if(temp_ss && (uint64_t)next_id + n > 0xffffffffull) die("read ids beyond 2^32 with temporary splice sites (use --no-temp-splicesite or split the input)");
		P.first_read_id = (uint32_t)next_id;
		sg.first_id = next_id;
		next_id += n;
		// the chain mode (window 0: waves of ONE read) is exact and meant for small inputs; an input that turns out not to be small is told so,
		// loudly and once (a small run's stderr stays the reference's summary, byte for byte)
		if(temp_ss && ss_window == 0 && next_id - skip >= 20000 && next_id - skip - n < 20000 && !getenv("H2G_QUIET_CHAIN_WARNING"))
			fprintf(stderr, "Warning: hisat2-align-amd: -p 1 with temporary splice sites is the reference's strict read-after-read chain (window 0, hisat2.cpp:3687): "
			                "it runs as waves of ONE read - a device round trip and a database merge per read; 20000 reads in, this input is not small. "
			                "Use -p >= 2 or --ss-window W (output == hisat2 -p W/1000 --reorder), or --no-temp-splicesite, for throughput.\n");
		if(paired) {
			if(h2g_set_mates(sg.st, b.codes.data(), b.offs.data(), b.have_quals ? b.quals.data() : nullptr, b.names.data(), b.noffs.data(), n) != H2G_OK) die("h2g_set_mates");
			if(h2g_align_pairs_run(sg.st, &P) != H2G_OK) die("h2g_align_pairs_run");
		} else if(h2g_align_run(sg.st, &P) != H2G_OK) die("h2g_align_run");
		t_up += now() - tq0;
		sg.batch = k; sg.n = n;
		t_gpu += now() - tg;
	}
	{   // drain, oldest first
		long oldest = -1;
		for(;;) {
			int gi = -1;
			for(int g = 0; g < G; g++) if(S[(size_t)g].batch >= 0 && (gi < 0 || S[(size_t)g].batch < oldest)) { gi = g; oldest = S[(size_t)g].batch; }
			if(gi < 0) break;
			const double tg = now();
			complete(gi);
			t_gpu += now() - tg;
		}
	}
	if(parser.joinable()) parser.join();
	ffinish();
	wfinish();
	if(werr) { fprintf(stderr, "Error: writing the SAM output failed\n"); return 1; }
	if(out != stdout) fclose(out); else fflush(out);
	if(!novel_out.empty()) {                              // hisat2.cpp:4189-4197
		FILE* nf = fopen(novel_out.c_str(), "w");
		if(nf) {
			const size_t need = h2g_sam_novel_splice_sites_text(sam, nullptr, 0);
			std::vector<char> tb(need + 1);
			h2g_sam_novel_splice_sites_text(sam, tb.data(), need);
			fwrite(tb.data(), 1, need, nf);
			fclose(nf);
		}
	}
	const double t2 = now();
	{   // the reference's alignment summary (aln_sink.h:1637), same text
		const size_t need = h2g_sam_summary(sam, nullptr, 0);
		std::vector<char> sb(need + 1);
		h2g_sam_summary(sam, sb.data(), need);
		if(!quiet) fwrite(sb.data(), 1, need, stderr);
		if(!quiet && !summary_file.empty()) { FILE* sf = fopen(summary_file.c_str(), "w"); if(sf) { fwrite(sb.data(), 1, need, sf); fclose(sf); } }   // hisat2.cpp:4175
	}
	(void)naligned; (void)nreads;
	// Reads whose lists overflow the default device workspace are re-run on the device with the large one (h2g_align_run's
	// second pass).  What is still flagged after that is NOT known to equal the reference's output: name it and fail.
	if(novf) fprintf(stderr, "Error: %llu %s exceeded even the large device workspace (h2g overflow bit); their SAM records are not verified "
	                 "against hisat2 -- rerun these with the reference aligner:\n%s", (unsigned long long)novf, paired ? "pairs" : "reads", ovf_names.c_str());
	if(getenv("H2G_CLI_TIMING")) fprintf(stderr, "time: index load %.2f s, align+fetch %.2f s (waited for the parser thread %.2f s; it parsed for %.2f s), SAM formatting %.2f s, total %.2f s [stream create %.2f, upload+launch %.2f, wait+fetch %.2f]\n", t1 - t0, t_gpu,
	        t_parse, t_parse_busy, t_fmt, t2 - t0, t_stream, t_up, t_fetch);
	if(!stats_fn.empty()) {
		FILE* sf = fopen(stats_fn.c_str(), "w");
		if(sf) { fprintf(sf, "{\"reads\": %llu, \"second_pass\": %llu, \"overflow\": %llu}\n", (unsigned long long)nreads, (unsigned long long)nsecond, (unsigned long long)novf); fclose(sf); }
	}
	// (Measured and not shipped, round 6: ending the process here without the frees below saves this run 0.1 s and costs the NEXT process 1.7 s — the driver reclaims 40 GB of
	// device memory of a process that did not return it while the next one is already allocating: profiles/r06_zc_ab.log.)
	for(auto& sg : S) if(sg.st) h2g_stream_free(sg.st);
	h2g_sam_close(sam);
	for(int g = 0; g < gpus; g++) { bool dup = false; for(int q = 0; q < g; q++) dup |= ixs[(size_t)q] == ixs[(size_t)g]; if(!dup) h2g_index_free(ixs[(size_t)g]); }
	return novf ? 3 : 0;
}
