// h2g_splice_host.h — host-side construction of the splice-site probability tables of SpliceSiteDB::probscore
// (splice_site.cpp:31-105, the model that is compiled in: NEW_PROB_MODEL is not defined).  The position weight matrices are
// the reference's model parameters (splice_site.cpp:31-44, background splice_site.h:66); the tables are derived from them
// exactly as init_junction_prob() derives them (same float expressions, same libm), uploaded once and read by hit_combine /
// calculate_score through DScoring.  tests/test_oracle_golden.py pins them against the reference's own tables.
#pragma once
#include <math.h>
#include <vector>

namespace h2g {

// log / exp of a float argument resolve to the float overloads in the reference (splice_site.h is `using namespace std`):
// logf / expf reproduce its tables bit for bit, the double versions do not (39 of 20 000 sampled entries differ)
#define H2G_SPL_LOG(x) logf(x)
#define H2G_SPL_EXP(x) expf(x)

inline void splice_tables(std::vector<float>& donor_sum, std::vector<float>& acc_sum1, std::vector<float>& acc_sum2) {
	const int donor_len = 9, acceptor_len = 15, acceptor_len1 = 7, acceptor_len2 = 8;
	float donor_prob[4][9] = {
		{0.340f, 0.604f, 0.092f, 0.001f, 0.001f, 0.526f, 0.713f, 0.071f, 0.160f},
		{0.363f, 0.129f, 0.033f, 0.001f, 0.001f, 0.028f, 0.076f, 0.055f, 0.165f},
		{0.183f, 0.125f, 0.803f, 1.000f, 0.001f, 0.419f, 0.118f, 0.814f, 0.209f},
		{0.114f, 0.142f, 0.073f, 0.001f, 1.000f, 0.025f, 0.093f, 0.059f, 0.462f}};
	float acceptor_prob[4][15] = {
		{0.090f, 0.084f, 0.075f, 0.068f, 0.076f, 0.080f, 0.097f, 0.092f, 0.076f, 0.078f, 0.237f, 0.042f, 1.000f, 0.001f, 0.239f},
		{0.310f, 0.310f, 0.307f, 0.293f, 0.326f, 0.330f, 0.373f, 0.385f, 0.410f, 0.352f, 0.309f, 0.708f, 0.001f, 0.001f, 0.138f},
		{0.125f, 0.115f, 0.106f, 0.104f, 0.110f, 0.113f, 0.113f, 0.085f, 0.066f, 0.064f, 0.212f, 0.003f, 0.001f, 1.000f, 0.520f},
		{0.463f, 0.440f, 0.470f, 0.494f, 0.471f, 0.463f, 0.408f, 0.429f, 0.445f, 0.504f, 0.240f, 0.246f, 0.001f, 0.001f, 0.104f}};
	const float background_prob[4] = {0.27f, 0.23f, 0.23f, 0.27f};
	for(int i = 0; i < donor_len; i++) for(int j = 0; j < 4; j++) donor_prob[j][i] = H2G_SPL_LOG(donor_prob[j][i] / background_prob[j]);
	for(int i = 0; i < acceptor_len; i++) for(int j = 0; j < 4; j++) acceptor_prob[j][i] = H2G_SPL_LOG(acceptor_prob[j][i] / background_prob[j]);
	donor_sum.assign((size_t)1 << (donor_len << 1), 0.0f);
	for(size_t i = 0; i < donor_sum.size(); i++) {
		float sum = 0.0f;
		for(int j = 0; j < donor_len; j++) sum += donor_prob[(i >> (j << 1)) & 3][donor_len - j - 1];
		donor_sum[i] = H2G_SPL_EXP(-sum);
	}
	acc_sum1.assign((size_t)1 << (acceptor_len1 << 1), 0.0f);
	for(size_t i = 0; i < acc_sum1.size(); i++) {
		float sum = 0.0f;
		for(int j = 0; j < acceptor_len1; j++) sum += acceptor_prob[(i >> (j << 1)) & 3][acceptor_len1 - j - 1];
		acc_sum1[i] = H2G_SPL_EXP(-sum);
	}
	acc_sum2.assign((size_t)1 << (acceptor_len2 << 1), 0.0f);
	for(size_t i = 0; i < acc_sum2.size(); i++) {
		float sum = 0.0f;
		for(int j = 0; j < acceptor_len2; j++) sum += acceptor_prob[(i >> (j << 1)) & 3][acceptor_len - j - 1];
		acc_sum2[i] = H2G_SPL_EXP(-sum);
	}
}

}  // namespace h2g
