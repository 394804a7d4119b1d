// msfl_odom.cuh — scan-to-scan association kernels (stage B). Filled in below.
#pragma once
