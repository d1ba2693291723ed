// l1_irls -- command-line driver with the arguments, file formats and defaults of the
// reference's demo binary (ral/test.cpp:75-333), written against the RAL-compatible shim
// (include/irotavg/l1_irls.hpp) and therefore running on the MI355X core.
//
//   l1_irls input_file [output_file [cost [sigma [irls_iters [l1_iters [change_th]]]]]]
//
// input : "m n f" / m lines "i j w x y z" (ids remapped to their sorted rank) / up to n lines
//         "w x y z" (at least f); output: n lines "w x y z" + m weights, 17 significant digits.
#include <cctype>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <set>

#include "../include/irotavg/l1_irls.hpp"

using namespace irotavg;

static Cost parse_cost(const char *name) {  // ral/test.cpp:35-72
    static const char *names[] = {"l2", "l1", "l1.5", "l0.5", "geman-mcclure", "huber",
                                  "pseudo-huber", "andrews", "bisquare", "cauchy", "fair",
                                  "logistic", "talwar", "welsch"};
    std::string s(name ? name : "");
    for (auto &c : s) c = (char)std::tolower((unsigned char)c);
    for (int k = 0; k < 14; k++)
        if (s == names[k]) return (Cost)k;
    std::cerr << "Unknown string. " << (name ? name : "") << std::endl;
    std::exit(-1);
}

int main(int argc, const char *argv[]) {
    const char *usage =
        "Usage:\n\nl1_irls input_file [output_file [cost [sigma [irls_iters [l1_iters [change_th]]]]]]\n\n"
        "  input_file   m n f / m x 'i j w x y z' / >= f x 'w x y z'\n"
        "  output_file  n x 'w x y z' then m IRLS weights (default l1_irls_out.txt)\n"
        "  cost         L2, L1, L1.5, L0.5, Geman-McClure, Huber, Pseudo-Huber, Andrews, Bisquare,\n"
        "               Cauchy, Fair, Logistic, Talwar, Welsch (default Geman-McClure)\n"
        "  sigma        IRLS sigma in degrees (default 5)\n"
        "  irls_iters   default 50;  l1_iters default 5;  change_th default 0.001\n";
    if (argc - 1 < 1 || argc - 1 > 7) {  // ral/test.cpp:138-155
        std::cerr << "Invalid number of arguments. Expected at least 1 and at most 7 arguments.\n"
                  << usage << std::endl;
        return 255;
    }
    std::ifstream in(argv[1]);
    if (!in.is_open()) {
        std::cerr << "Unable to open file " << argv[1] << std::endl;
        return 255;
    }
    int m, n, f;
    in >> m >> n >> f;
    std::cout << "# rel rots ..... = " << m << "\n# abs rots ..... = " << n
              << "\n# fixed abs rots = " << f << std::endl;
    I_t I;
    I.reserve((size_t)m);
    Mat QQ = Mat::Zero(m, 4), Q = Mat::Zero(n, 4);
    std::set<int> vertices;
    for (int k = 0; k < m; k++) {  // :180-200 -- file order w x y z, memory order x y z w
        int e1, e2;
        double w, x, y, z;
        if (!(in >> e1 >> e2 >> w >> x >> y >> z)) {
            std::cerr << "Corrupt input file: inconsistent number of connections." << std::endl;
            return 255;
        }
        I.push_back(std::make_pair(e1, e2));
        vertices.insert(e1);
        vertices.insert(e2);
        QQ(k, 0) = x; QQ(k, 1) = y; QQ(k, 2) = z; QQ(k, 3) = w;
    }
    std::map<int, int> v2i;  // :202-213
    int next = 0;
    for (int v : vertices) v2i[v] = next++;
    for (auto &c : I) {
        c.first = v2i[c.first];
        c.second = v2i[c.second];
    }
    int i = 0;
    while (i < n) {  // :216-228
        double w, x, y, z;
        if (!(in >> w >> x >> y >> z)) break;
        Q(i, 0) = x; Q(i, 1) = y; Q(i, 2) = z; Q(i, 3) = w;
        i++;
    }
    if (i < f) {
        std::cerr << "Insuficient number of absolute rotations. At least " << f << " must be given." << std::endl;
        return 255;
    }
    int maxj = -1;
    for (auto &e : I) maxj = std::max(maxj, e.second);
    if (n != maxj + 1) {  // :236-247
        std::cerr << "Corrupt input file: check abs rotations" << std::endl;
        return 255;
    }
    const char *output_file = (argc - 1 > 1) ? argv[2] : "l1_irls_out.txt";
    const Cost cost = (argc - 1 > 2) ? parse_cost(argv[3]) : Geman_McClure;
    const double sigma = ((argc - 1 > 3) ? std::atof(argv[4]) : 5.0) * M_PI / 180.0;
    const int irls_iters = (argc - 1 > 4) ? std::atoi(argv[5]) : 50;
    const int l1_iters = (argc - 1 > 5) ? std::atoi(argv[6]) : 5;
    const double change_th = (argc - 1 > 6) ? std::atof(argv[7]) : 1e-3;
    std::cout << "output file: " << output_file << "\ncost: " << cost << "\nsigma [deg]: "
              << sigma * 180. / M_PI << "\nIRLS max. iterations: " << irls_iters
              << "\nL1-RA max. iterations: " << l1_iters << "\nchange threshold: " << change_th << std::endl;
    if (f == 0) {  // :277-282
        Q(0, 0) = 0; Q(0, 1) = 0; Q(0, 2) = 0; Q(0, 3) = 1;
        std::cout << "set first abs rot = I" << std::endl;
        f = 1;
    }
    const int init_f = (i > f) ? i : f;  // :284-286
    init_mst(Q, QQ, I, init_f);
    SpMat A = make_A(n, f, I);
    int l1_out = 0, irls_out = 0;
    double l1_rt = 0, irls_rt = 0;
    l1ra(QQ, I, A, Q, f, l1_iters, change_th, l1_out, l1_rt);
    Vec weights(m);
    irls(QQ, I, A, cost, sigma, Q, f, irls_iters, change_th, weights, irls_out, irls_rt);
    quat_normalised(Q, f);
    std::cout << "L1-RA iterations = " << l1_out << "\nIRLS  iterations = " << irls_out
              << "\nL1-RA runtime [s] = " << l1_rt << "\nIRLS  runtime [s] = " << irls_rt
              << "\ntotal runtime [s] = " << (l1_rt + irls_rt) << std::endl;
    std::ofstream out(output_file);
    if (!out.is_open()) {
        std::cerr << "Unable to save results." << std::endl;
        return 1;
    }
    out << std::setprecision(17);
    for (int r = 0; r < n; r++) out << Q(r, 3) << " " << Q(r, 0) << " " << Q(r, 1) << " " << Q(r, 2) << "\n";
    for (int k = 0; k < m; k++) out << weights(k) << "\n";
    return 0;
}
