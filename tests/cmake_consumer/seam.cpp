// Links the kernel boundary through the tvlqr::tvlqr target: workspace size of a 10-step (4, 2) problem.
#include <cstdio>
#include <vector>

#include "tvlqr/tvlqr.h"

int main() {
  std::vector<int> nx(11, 4), nu(10, 2);
  const int bytes = tvlqr_TotalMemSize(nx.data(), nu.data(), 10, false);
  std::printf("tvlqr_TotalMemSize = %d\n", bytes);
  return bytes > 0 ? 0 : 1;
}
