// exceptions.hpp -- error codes of the altro::ALTROSolver API and how they are reported.
//
// The enumerator names and their ORDER are API (callers compare against ErrorCodes::NoError, tests check
// specific codes; reference: src/altro/solver/exceptions.hpp:24-51), so they are generated from one table that
// also carries the message text.  Reporting follows the reference's ALTRO_THROW convention
// (exceptions.hpp:13-20): by default the error is printed in red on stderr and the code is RETURNED;
// with ALTRO_ENABLE_RUNTIME_EXCEPTIONS defined it is thrown as AltroErrorException.
#pragma once

#include <cstdio>
#include <stdexcept>
#include <string>

// X(code, message) -- one row per error, in API order.
#define ALTRO_ERROR_TABLE(X)                                                                             \
  X(NoError, "no error")                                                                                 \
  X(StateDimUnknown, "state dimension unknown")                                                          \
  X(InputDimUnknown, "input dimension unknown")                                                          \
  X(NextStateDimUnknown, "next state dimension unknown")                                                 \
  X(DimensionUnknown, "dimension unknown")                                                               \
  X(BadIndex, "bad knot point index")                                                                    \
  X(DimensionMismatch, "dimension mismatch")                                                             \
  X(SolverNotInitialized, "solver not initialized")                                                      \
  X(SolverAlreadyInitialized, "solver already initialized")                                              \
  X(NonPositive, "expected a positive value")                                                            \
  X(TimestepNotPositive, "time step not positive")                                                       \
  X(CostFunNotSet, "cost function not set")                                                              \
  X(DynamicsFunNotSet, "dynamics function not set")                                                      \
  X(InvalidOptAtTerminalKnotPoint, "invalid operation at the terminal knot point")                       \
  X(MaxConstraintsExceeded, "maximum number of constraints exceeded")                                    \
  X(InvalidConstraintDim, "invalid constraint dimension")                                                \
  X(CholeskyFailed, "Cholesky factorization failed")                                                     \
  X(OpOnlyValidAtTerminalKnotPoint, "operation only valid at the terminal knot point")                   \
  X(InvalidPointer, "invalid pointer")                                                                   \
  X(BackwardPassFailed, "backward pass failed (try increasing regularization)")                          \
  X(LineSearchFailed, "line search failed to find a point satisfying the strong Wolfe conditions")       \
  X(MeritFunctionGradientTooSmall, "merit function gradient under tolerance")                            \
  X(InvalidBoundConstraint, "invalid bound constraint")                                                  \
  X(NonPositivePenalty, "penalty must be strictly positive")                                             \
  X(CostNotQuadratic, "cost function not quadratic")                                                     \
  X(FileError, "file error")

namespace altro {

enum class ErrorCodes {
#define ALTRO_ERROR_ENUMERATOR(code, message) code,
  ALTRO_ERROR_TABLE(ALTRO_ERROR_ENUMERATOR)
#undef ALTRO_ERROR_ENUMERATOR
};

const char* ErrorCodeToString(ErrorCodes err);  // message column of the table ("unknown error" out of range)
void PrintErrorCode(ErrorCodes err);            // "Got error code <int>: <message>" on stderr

class AltroErrorException : public std::runtime_error {
 public:
  AltroErrorException(std::string msg, ErrorCodes code) : std::runtime_error(msg.c_str()), code_(code) {}
  virtual ~AltroErrorException() {}
  virtual ErrorCodes Errno() { return code_; }

 private:
  ErrorCodes code_;
};

namespace detail {
inline ErrorCodes Report(const std::string& msg, ErrorCodes code, const char* file, int line) {
  std::fprintf(stderr, "\033[31mALTRO ERROR Code %d: %s %s:%d\n  Message: %s\033[0m\n", static_cast<int>(code),
               ErrorCodeToString(code), file, line, msg.c_str());
  return code;
}
}  // namespace detail

}  // namespace altro

#undef ALTRO_THROW
#ifdef ALTRO_ENABLE_RUNTIME_EXCEPTIONS
#define ALTRO_THROW(msg, code) (throw(::altro::AltroErrorException((msg), code)), code)
#else
#define ALTRO_THROW(msg, code) (::altro::detail::Report((msg), (code), __FILE__, __LINE__))
#endif
