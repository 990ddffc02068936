// main.cc -- ./step50_mi355x <file.prm>: the counterpart of the reference's src/main.cc:6-121.
#include <iostream>

#include "laplace_problem.h"

int main(int argc, char *argv[]) {
  using namespace step50;
  try {
    if (argc != 2) {
      std::cerr << "usage: " << argv[0] << " <parameter file .prm>" << std::endl;
      return 1;
    }
    ParameterReader prm;
    prm.declare_parameters();
    prm.read_parameters(argv[1]);
    const Parameters par = Parameters::from(prm);
    if (par.dim == 2) {
      LaplaceProblem<2> problem(par);
      problem.echo = true;
      std::cout << problem.log;
      problem.run();
    } else if (par.dim == 3) {
      LaplaceProblem<3> problem(par);
      problem.echo = true;
      std::cout << problem.log;
      problem.run();
    } else {
      throw std::runtime_error("Only 2d and 3d dimensions are supported.");
    }
  } catch (std::exception &exc) {  // src/main.cc:96-118
    std::cerr << std::endl << "----------------------------------------------------" << std::endl;
    std::cerr << "Exception on processing: " << std::endl << exc.what() << std::endl << "Aborting!" << std::endl;
    return 1;
  }
  return 0;
}
