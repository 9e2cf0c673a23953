// cli_common.hpp -- what singlePopMcmc / multiPopMcmc / makeCMD share: settings -> context.
#pragma once
#include "b9host.hpp"

namespace b9h {

struct Session {
    Settings settings;
    ModelPack pack;
    Photometry phot;
    b9_ctx *ctx = nullptr;
    std::vector<double> start;        // B9_NPARAM starting row
    b9_priors priors{};
    McmcConfig mcmc;
    std::string output_base;
    ~Session() { if (ctx) b9_ctx_destroy(ctx); }
};

// Parses flags / YAML, loads the photometry (unless `need_phot` is false) and the model pack with
// the photometry's filters, creates the GPU context and stages everything.  Throws on error.
void open_session(Session &s, int argc, char **argv, int n_pops, bool need_phot);

int report_and_exit_code(const char *prog, const std::exception &e);

}  // namespace b9h
