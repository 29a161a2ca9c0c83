/* GASAL2/include/args_parser.h -- see gasal.h in this directory. */
#ifndef ARGS_PARSER_H
#define ARGS_PARSER_H
#include "gasal.h"

class Parameters {
public:
	Parameters(int argc_, char **argv_);
	~Parameters();
	void print();
	void failure(int f);
	void help();
	void parse();
	void fileopen();

	int32_t sa, sb, gapo, gape;
	comp_start start_pos;
	int print_out;
	int n_threads;
	int32_t k_band;
	bool secondBest;
	bool isPacked;
	bool isReverseComplement;
	data_source semiglobal_skipping_head;
	data_source semiglobal_skipping_tail;
	algo_type algo;
	int argc;
	char **argv;
};
#endif
