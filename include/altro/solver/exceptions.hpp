// exceptions.hpp -- error codes and the ALTRO_THROW convention of the reference
// (src/altro/solver/exceptions.hpp:13-68): by default an error is printed (in red) and the code is
// RETURNED; with ALTRO_ENABLE_RUNTIME_EXCEPTIONS defined it is thrown as AltroErrorException.
// Enumerator order is part of the API (callers compare against ErrorCodes::NoError etc.).
#pragma once

#include <cstdio>
#include <stdexcept>
#include <string>

namespace altro {

enum class ErrorCodes {
  NoError,
  StateDimUnknown,
  InputDimUnknown,
  NextStateDimUnknown,
  DimensionUnknown,
  BadIndex,
  DimensionMismatch,
  SolverNotInitialized,
  SolverAlreadyInitialized,
  NonPositive,
  TimestepNotPositive,
  CostFunNotSet,
  DynamicsFunNotSet,
  InvalidOptAtTerminalKnotPoint,
  MaxConstraintsExceeded,
  InvalidConstraintDim,
  CholeskyFailed,
  OpOnlyValidAtTerminalKnotPoint,
  InvalidPointer,
  BackwardPassFailed,
  LineSearchFailed,
  MeritFunctionGradientTooSmall,
  InvalidBoundConstraint,
  NonPositivePenalty,
  CostNotQuadratic,
  FileError,
};

void PrintErrorCode(ErrorCodes err);
const char* ErrorCodeToString(ErrorCodes err);

class AltroErrorException : public std::runtime_error {
 public:
  AltroErrorException(std::string msg, ErrorCodes code) : std::runtime_error(msg.c_str()), code_(code) {}
  virtual ErrorCodes Errno() { return code_; }
  virtual ~AltroErrorException() {}

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
