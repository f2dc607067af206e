// `mara_hip <sub-program> key=value...` - dispatch as in the reference's src/app_main.cpp:53-95
// (registry built by hand, total execution time printed, exceptions end the run).
#include <chrono>
#include <cstdio>
#include <map>
#include "app_subprogram.hpp"

int main(int argc, const char* argv[])
{
    std::map<std::string, std::unique_ptr<mara::sub_program_t>> programs;
    programs["sedov"] = make_subprog_sedov();
    programs["euler2d"] = make_subprog_euler2d();
    programs["cloud"] = make_subprog_cloud();
    programs["binary"] = make_subprog_binary();

    if (argc == 1)
    {
        std::printf("usages: \n");
        for (auto& prog : programs) std::printf("    mara_hip %s\n", prog.first.c_str());
        return 0;
    }
    if (programs.count(argv[1]))
    {
        const auto t0 = std::chrono::high_resolution_clock::now();
        int code = 1;
        try
        {
            code = programs.at(argv[1])->main(argc - 1, argv + 1);
        }
        catch (const std::exception& e)
        {
            std::printf("%s\n", e.what());
            return 1;
        }
        std::printf("total execution time: %g seconds\n", std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count());
        return code;
    }
    std::printf("invalid sub-program '%s'\n", argv[1]);
    return 0;
}
