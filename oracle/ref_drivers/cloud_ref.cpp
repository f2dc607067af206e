/**
 * TEST INFRASTRUCTURE — reference-side oracle driver for BASELINE config 4
 * (`mara cloud`): 2-D axisymmetric spherical-polar SRHD jet-cloud.
 *
 * The sub-program itself needs HDF5 (subprog_cloud.cpp includes core_hdf5.hpp),
 * so this driver calls the same reference header functions, in the same order,
 * as CloudProblem::new_solution (subprog_cloud.cpp:610-662), the geometry
 * helpers (:260-290), the boundary conditions (:466-509), advance (:511-584)
 * and next_solution (:676-697), with gamma = 4/3 (:52) and the option defaults
 * of :60-87.
 *
 * usage: cloud_ref <nr> <num_decades> <rk_order> <reconstruct_method> <plm_theta> <nsteps> <outprefix>
 * writes <outprefix>.{rv,qv,u0,un,inflow,meta,diag_fields,diag_columns,diag_meta}.f64
 *   inflow: [nsteps][nq][5] primitives of the inner radial ghost cells at each step's start time
 *   meta  : dt, temperature_floor
 *   diag_*: CloudProblem::make_diagnostic_fields (subprog_cloud.cpp:334-433) of the final state, composed from the same header
 *           functions (post_shock_locator.hpp, nd::freeze_axis, ...): diag_fields [5][nr][nq] = mass_density, gas_pressure,
 *           specific_entropy, radial_gamma_beta, radial_energy_flow; diag_columns [15][nq] in the order of diagnostic_fields_t
 *           (:147-161: total_energy, solid_angle, the four shock radii, postshock gamma, power, power02..64, power_max);
 *           diag_meta = time (s), reference length, mass, time
 */
#include "cloud_compose.hpp"
using namespace cloud_compose;

int main(int argc, char** argv)
{
    if (argc != 8 && argc != 9) return 1;
    // optional 8th argument: threads = 1, 2, 4, 8, 12, 16 or 32 - the evaluator upstream uses (mara::evaluate_on<threads>(), src/app_parallel.hpp:72-103)
    // in the place of nd::to_shared(); the bits do not depend on it. bench_configs.py times the reference's own threaded CPU path with it.
    const int threads = argc == 9 ? std::atoi(argv[8]) : 0;
    int nr = std::atoi(argv[1]);
    double num_decades = std::atof(argv[2]);
    int rk = std::atoi(argv[3]);
    setup_t S;
    S.method = std::atoi(argv[4]);
    S.theta = std::atof(argv[5]);
    int nsteps = std::atoi(argv[6]);
    std::string prefix = argv[7];

    double ref_length = 0.0, ref_mass = 0.0;
    auto u = make_cloud_problem(S, nr, num_decades, ref_length, ref_mass);
    auto u_init = u;
    const double cfl_number = 0.4;          // subprog_cloud.cpp:60-87

    auto dr_min = S.rv | nd::difference_on_axis(0) | nd::read_index(0);
    auto dt = dr_min / mara::make_velocity(1.0) * cfl_number;
    double time = 0.0;
    std::vector<double> inflow;

    for (int n = 0; n < nsteps; ++n)
    {
        u = with_upstream_evaluator(threads, [&] (auto evaluate)
        {
            if (rk == 1) return advance(S, u, time, dt, &inflow, evaluate);
            auto s1 = advance(S, u, time, dt, &inflow, evaluate);
            auto s2 = advance(S, s1, time, dt, nullptr, evaluate);         // both stages see the step-start time (:468-473, :524)
            return (u * 0.5 + s2 * (1 - 0.5)) | nd::to_shared();
        });
        time += dt.value;
    }

    auto dump = [&] (const char* ext, const void* data, std::size_t bytes)
    {
        FILE* f = std::fopen((prefix + "." + ext + ".f64").c_str(), "wb");
        std::fwrite(data, 1, bytes, f);
        std::fclose(f);
    };
    static_assert(sizeof(mara::srhd::conserved_t) == 40, "layout");
    dump("rv", S.rv.data(), S.rv.size() * sizeof(double));
    dump("qv", S.qv.data(), S.qv.size() * sizeof(double));
    dump("u0", u_init.data(), u_init.size() * 40);
    dump("un", u.data(), u.size() * 40);
    dump("inflow", inflow.data(), inflow.size() * sizeof(double));
    double meta[2] = {dt.value, S.temperature_floor};
    dump("meta", meta, sizeof meta);

    // ---- make_diagnostic_fields, subprog_cloud.cpp:334-433 (unit_system_t :177-195)
    {
        using namespace std::placeholders;
        const double u_length = ref_length, u_mass = ref_mass, u_time = S.ref_time;
        const double u_energy = u_mass * std::pow(light_speed_cgs, 2);
        const double u_mass_density = u_mass / std::pow(u_length, 3);
        const double u_energy_density = u_energy / std::pow(u_length, 3);
        const double u_power = u_energy / u_time;
        auto dv           = cell_volumes(S.rv, S.qv);
        auto dAr          = radial_face_areas(S.rv, S.qv);
        auto rhat         = mara::unit_vector_t::on_axis_1();
        auto cons_to_prim = std::bind(mara::srhd::recover_primitive, _1, gamma_law, S.temperature_floor);
        auto radial_cells = S.rv | nd::midpoint_on_axis(0);
        auto primitive    = u | nd::divide(dv) | nd::map(cons_to_prim);
        const std::size_t nq = S.qv.size() - 1, nrr = S.rv.size() - 1;
        std::vector<std::vector<double>> col(15, std::vector<double>(nq));
        for (std::size_t j = 0; j < nq; ++j)
        {
            auto pj = primitive | nd::freeze_axis(1).at_index(j) | nd::to_shared();
            auto uj = u         | nd::freeze_axis(1).at_index(j);
            auto Aj = dAr | nd::freeze_axis(1).at_index(j) | nd::midpoint_on_axis(0);
            auto Lj = pj | nd::map([rhat] (auto p) { return p.flux(rhat, gamma_law)[4]; })
            | nd::multiply(Aj)
            | nd::multiply(u_power)
            | nd::map([] (auto L) { return L.value; });

            auto midpoint_index   = mara::find_shock_index(pj, gamma_law)[0];
            auto upstream_index   = mara::find_index_of_pressure_plateau_ahead(pj, midpoint_index);
            auto pressure_index   = mara::find_index_of_maximum_pressure_behind(pj, midpoint_index);
            auto luminosity_index = mara::find_index_of_maximum_behind(Lj, midpoint_index);
            auto i02 = midpoint_index >  2 ? midpoint_index -  2 : 0;
            auto i04 = midpoint_index >  4 ? midpoint_index -  4 : 0;
            auto i08 = midpoint_index >  8 ? midpoint_index -  8 : 0;
            auto i16 = midpoint_index > 16 ? midpoint_index - 16 : 0;
            auto i32 = midpoint_index > 32 ? midpoint_index - 32 : 0;
            auto i64 = midpoint_index > 64 ? midpoint_index - 64 : 0;

            col[0][j]  = uj | nd::map([] (auto u_) { return u_[4].value; }) | nd::multiply(u_energy) | nd::sum();
            col[1][j]  = dAr(0, j) / S.rv(0) / S.rv(0);
            col[2][j]  = radial_cells(midpoint_index).value * u_length;
            col[3][j]  = radial_cells(upstream_index).value * u_length;
            col[4][j]  = radial_cells(pressure_index).value * u_length;
            col[5][j]  = radial_cells(luminosity_index).value * u_length;
            col[6][j]  = primitive(pressure_index, j).lorentz_factor();
            col[7][j]  = Lj(pressure_index);
            col[8][j]  = Lj(i02);
            col[9][j]  = Lj(i04);
            col[10][j] = Lj(i08);
            col[11][j] = Lj(i16);
            col[12][j] = Lj(i32);
            col[13][j] = Lj(i64);
            col[14][j] = Lj(luminosity_index);
        }
        auto specific_entropy   = primitive | nd::map(std::bind(&prim_t::specific_entropy, _1, gamma_law)) | nd::to_shared();
        auto gas_pressure       = primitive | nd::map(std::mem_fn(&prim_t::gas_pressure)) | nd::multiply(u_energy_density) | nd::to_shared();
        auto mass_density       = primitive | nd::map(std::mem_fn(&prim_t::mass_density)) | nd::multiply(u_mass_density) | nd::to_shared();
        auto radial_gamma_beta  = primitive | nd::map(std::mem_fn(&prim_t::gamma_beta_1)) | nd::to_shared();
        auto radial_energy_flow = primitive
        | nd::map([rhat] (auto p) { return p.flux(rhat, gamma_law); })
        | nd::multiply(dAr | nd::select_axis(0).from(0).to(1).from_the_end())
        | nd::map([] (auto L) { return L[4].value; })
        | nd::multiply(u_power)
        | nd::to_shared();
        std::vector<double> fields;
        for (auto v : mass_density) fields.push_back(v);
        for (auto v : gas_pressure) fields.push_back(v);
        for (auto v : specific_entropy) fields.push_back(v);
        for (auto v : radial_gamma_beta) fields.push_back(v);
        for (auto v : radial_energy_flow) fields.push_back(v);
        dump("diag_fields", fields.data(), fields.size() * sizeof(double));
        std::vector<double> columns;
        for (auto& c : col) for (double v : c) columns.push_back(v);
        dump("diag_columns", columns.data(), columns.size() * sizeof(double));
        double dmeta[4] = {time * u_time, u_length, u_mass, u_time};
        dump("diag_meta", dmeta, sizeof dmeta);
        (void) nrr;
    }
    return 0;
}
