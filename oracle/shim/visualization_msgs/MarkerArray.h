#pragma once
#include <visualization_msgs/Marker.h>
